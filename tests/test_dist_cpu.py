"""world_size-2 run of the multi-GPU exchange (localhgt_amd/dist.py) on CPU with gloo: the packed
saturating reduce-scatter/all-gather of the count table and the vote all-reduce.  The device side
is replaced by a numpy adapter here; the GPU adapter is exercised by tests/test_gpu_dist.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def unpack(packed: np.ndarray) -> np.ndarray:
    out = np.empty(packed.size * 4, dtype=np.uint8)
    for f in range(4):
        out[f::4] = (packed >> (2 * f)) & 3
    return out


def pack(fields: np.ndarray) -> np.ndarray:
    f = fields.reshape(-1, 4).astype(np.uint8)
    return (f[:, 0] | (f[:, 1] << 2) | (f[:, 2] << 4) | (f[:, 3] << 6)).astype(np.uint8)


class FakeEngine:
    def __init__(self, table_u8: np.ndarray, votes: np.ndarray):
        self.table = torch.from_numpy(pack(table_u8))
        self.votes = torch.from_numpy(votes.astype(np.int32))


class NumpyAdapter:
    def counts_tensor(self, eng):
        return eng.table

    def merge(self, eng, other, byte_offset):
        mine = eng.table[byte_offset:byte_offset + other.numel()]
        merged = np.minimum(3, unpack(mine.numpy()).astype(int) + unpack(other.numpy()).astype(int)).astype(np.uint8)
        mine.copy_(torch.from_numpy(pack(merged)))

    def filter_tensor(self, eng):
        return eng.votes

    def sync(self, eng):
        pass

    # reference-sharded phase B: a fake rank "finds" eng.n_new peaks and 2 registrations per peak
    def scan_local(self, eng, hit_ratio, match_ratio):
        return eng.n_new, 3 * eng.n_new

    def scan_emit(self, eng, id_base, n_new):
        ids = torch.arange(id_base, id_base + n_new, dtype=torch.int32)
        loci = torch.stack([torch.full((n_new,), eng.rank + 1, dtype=torch.int32), ids * 10], dim=1).reshape(-1)
        regs = torch.stack([ids * 7 + 1, ids, ids * 7 + 2, ids], dim=1).reshape(-1).to(torch.int32)
        return loci, regs

    def peaks_install(self, eng, n_total, n_sel_total, max_peak, loci_all, regs_all):
        eng.installed = (n_total, n_sel_total, loci_all.clone(), regs_all.clone())


def _worker(rank, world, port, n_bytes, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from localhgt_amd.dist import Exchange
    ex = Exchange.from_env(backend="gloo", adapter=NumpyAdapter())
    rng = np.random.default_rng(100 + rank)
    table = rng.choice(4, size=n_bytes * 4, p=[.6, .2, .1, .1]).astype(np.uint8)
    votes = rng.integers(0, 300, size=1000)
    eng = FakeEngine(table, votes)
    assert ex.broadcast_flag(rank == 0) is True
    ex.merge_counts(eng)
    ex.sum_votes(eng)
    # variable-length all-gather, including an empty contribution
    got = ex.all_gather_var(torch.arange(5 if rank == 0 else 0, dtype=torch.int32) + 100 * rank)
    assert got.tolist() == [0, 1, 2, 3, 4]
    got = ex.all_gather_var(torch.arange(2 + 3 * rank, dtype=torch.int32) + 100 * rank)
    assert got.tolist() == [100 * r + i for r in range(world) for i in range(2 + 3 * r)]
    # sharded scan: rank 0 finds 4 peaks, rank 1 finds 0 (empty shard) or 3
    for n1 in (0, 3):
        eng.rank, eng.n_new = rank, (4 if rank == 0 else n1 if rank == 1 else 0)
        assert ex.sharded_scan(eng, 0.1, 0.08, 1000) == 4 + n1
        n_total, n_sel, loci, regs = eng.installed
        assert (n_total, n_sel) == (4 + n1, 3 * (4 + n1))
        ids = list(range(4 + n1))
        assert loci.view(-1, 2)[:, 1].tolist() == [10 * i for i in ids]               # ids follow rank (= contig) order
        assert loci.view(-1, 2)[:, 0].tolist() == [1] * 4 + [2] * n1
        assert sorted(regs.view(-1, 2)[:, 1].tolist()) == sorted(ids + ids)
    np.save(os.path.join(tmp, f"table_in_{rank}.npy"), table)
    np.save(os.path.join(tmp, f"votes_in_{rank}.npy"), votes)
    np.save(os.path.join(tmp, f"table_out_{rank}.npy"), unpack(eng.table.numpy()))
    np.save(os.path.join(tmp, f"votes_out_{rank}.npy"), eng.votes.numpy())
    ex.barrier()
    ex.close()


@pytest.mark.parametrize("world,n_bytes", [(2, 1 << 16), (3, 1 << 16), (3, 40)])
def test_exchange_world2_gloo(tmp_path, world, n_bytes):
    """world 3: the packed table does not split into equal word-aligned slices"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n_bytes, str(tmp_path)), nprocs=world, join=True)
    tin = [np.load(tmp_path / f"table_in_{r}.npy") for r in range(world)]
    vin = [np.load(tmp_path / f"votes_in_{r}.npy") for r in range(world)]
    want_t = np.minimum(3, sum(t.astype(int) for t in tin))
    want_v = sum(vin)
    for r in range(world):
        assert (np.load(tmp_path / f"table_out_{r}.npy") == want_t).all()
        assert (np.load(tmp_path / f"votes_out_{r}.npy") == want_v).all()


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts its two ranks itself; --dry-run drives the launcher, the process
    group and every exchange on host tensors (gloo) and prints the one JSON line on rank 0"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["world_size"] == 2 and d["exchanges_consistent"]
    # under a launcher (WORLD_SIZE set) it is one of the ranks and must not start more
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=1" in (res.stderr + res.stdout)


def test_saturating_sum_identity():
    """min(3, sum_r min(3, c_r)) == min(3, sum_r c_r): why per-rank saturated tables can be merged"""
    rng = np.random.default_rng(0)
    c = rng.integers(0, 9, size=(8, 10000))
    assert (np.minimum(3, np.minimum(3, c).sum(0)) == np.minimum(3, c.sum(0))).all()


def test_bench_helpers_planted_transfers_and_traffic_stamps(tmp_path):
    """bench.py's host-side helpers: the breakpoints the synthetic sample carries (the same hash as k_synth.hip: transfer_sites),
    interval recall as the reference's evaluation defines it, and the rule that a committed traffic figure is only used while the
    kernel sources it was measured on are unchanged"""
    import importlib
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    bp = bench.planted_breakpoints(1000, 1_000_000)
    assert len(bp) == 750 and bp[:3] == [(1, 953947), (2, 109604), (2, 112604)]          # values of the device generator (seed 1)
    assert all(1 <= c <= 500 and 3000 <= p < 1_000_000 for c, p in bp)
    assert len(bench.planted_breakpoints(13000, 1_000_000, 300)) == 450
    iv = tmp_path / "iv.txt"
    iv.write_text("1\t1\t1\n1\t953000\t954000\n2\t-100\t900\n2\t109000\t110000\n")
    rec = bench.interval_recall(str(iv), bp)
    assert rec["inside_an_interval"] == 2 and rec["breakpoints"] == 750 and rec["interval_lines"] == 4
    # traffic fallback: entries whose stamp matches the sources are fresh, the others are ignored
    stamps = {ph: bench.source_stamp(srcs) for ph, srcs in bench.KERNEL_SOURCES.items()}
    fake = {"tagA": {"count_A": {"bytes": 1}, "ref_flags": {"bytes": 2}, "vote_kernel": {"bytes": 3},
                     "_stamp": dict(stamps, ref_flags="0000000000000000")}}
    path = os.path.join(root, "profiles", "traffic_per_launch.json")
    real = open(path).read()
    try:
        open(path, "w").write(json.dumps(fake))
        fresh, stale = bench.committed_traffic("tagA")
        assert set(fresh) == {"count_A", "vote_kernel"} and set(stale) == {"ref_flags"}
        assert bench.committed_traffic("other") == ({}, {})
    finally:
        open(path, "w").write(real)
    committed = json.loads(real)
    assert all("_stamp" in v for v in committed.values())
