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


def _backend_worker(rank, world, port, tmp):
    """two ranks on a node with ONE GPU, started without LOCAL_WORLD_SIZE (srun / mpirun / by hand): rank 0 alone sees nothing
    wrong (LOCAL_RANK 0 < 1 GPU), rank 1 does -- both must end up staging through gloo (they picked nccl and gloo)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LHGT_DIST_INIT_BACKEND="gloo")
    for var in ("LOCAL_WORLD_SIZE", "LHGT_DIST_BACKEND", "SLURM_NTASKS_PER_NODE", "OMPI_COMM_WORLD_LOCAL_SIZE"):
        os.environ.pop(var, None)
    torch.cuda.device_count = lambda: 1               # the node's one GPU
    torch.cuda.set_device = lambda d: None
    from localhgt_amd.dist import Exchange
    ex = Exchange.from_env(adapter=NumpyAdapter())
    assert ex.agree(rank) == world - 1                # the group works
    open(os.path.join(tmp, f"backend_{rank}.txt"), "w").write(f"{ex.backend} {ex.device}")
    ex.close()


def test_ranks_agree_on_the_backend_when_only_one_of_them_sees_the_shared_gpu(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_backend_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [open(tmp_path / f"backend_{r}.txt").read() for r in range(2)]
    assert got == ["gloo 0", "gloo 0"], got


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
    # the N > 1 compact line: what the driver and a reader of a scaling run need survives compaction
    assert len(lines[0]) < 4096
    assert set(d["exchange_ms"]) == {"merge_counts", "sharded_scan", "sum_votes"} and "n1_equivalent_ms" in d
    assert "sharded_index" in d and d["scaling"] == "weak" and d["config"]["parallelism"] == "reads sharded x2"
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


def _canned_detail():
    """a full bench record with every optional part present and long strings where the real run has long strings"""
    roof = {"kernel": "vote_kernel_queued", "what": "x" * 300, "bound": "hbm", "peak": 8000.0, "unit": "GB/s", "ms_per_step": 383.476, "launches_per_step": 6,
            "launch_ms": 63.913, "bytes_model": 4577100000000, "frac_model": 1.492, "model_exceeded": True, "bytes_needed": 155000000000, "needed_is": "y" * 400,
            "frac_needed": 0.0505, "achieved": 2511.46, "frac": 0.3139, "traffic": 160514011050, "traffic_per_step": 963084066304, "overfetch": 6.21,
            "frac_raw": 0.157, "request_rate": {"value": 204.5, "unit": "G requests/s", "counter": "TCP_TCC_READ_REQ_sum", "ceiling": 254.0, "frac_of_ceiling": 0.805,
                                                "ceiling_source": "z" * 200}, "l2_hit_rate": 0.9041, "traffic_source": "w" * 150}
    leg = {"value": 134.168, "unit": "M paired-reads/s", "ms_per_step": 745.34, "phase_ms": {"count_A": 283.6, "scan_B": 369.27, "vote_C": 91.66},
           "scan_B_form": {"lite": False, "form": "trio-first"}, "raw_peaks": 2625, "filtered_peaks": 1307, "steps": 3, "pairs": 100000000,
           "planted_transfers": {"breakpoints": 450, "inside_an_interval": 450, "recall": 1.0, "interval_lines": 400}, "roofline": dict(roof, kernel="ref_flags_trio"),
           "roofline_other": {"count_A": roof, "vote_kernel": roof}, "workload": "v" * 200}
    return {"metric": "M paired-reads/s k-mer sketch->peak, UHGG-scale ref; %HBM roofline @1/2/4/8 GPU", "value": 95.7502, "unit": "M paired-reads/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 1044.385, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 13000x1000000 bp synthetic ref (13.00 Gbase, index resident), 100000000 150bp pairs per GPU drawn from half of its contigs, k=32 e=3, sample=1, phases A-D",
                       "pairs_per_gpu": 100000000, "ref_bases": 13000000000, "k": 32, "e": 3, "parallelism": "reads sharded x1"},
            "world_size": 1, "backend": None, "phase_ms": {"count_A": 285.451, "scan_B": 374.89, "vote_C": 383.476}, "exchange_ms": None, "n1_equivalent_ms": 1044.385,
            "raw_peaks": 25119, "filtered_peaks": 0, "planted_transfers": {"breakpoints": 9750, "inside_an_interval": 0, "recall": 0.0, "interval_lines": 1},
            "verify": {"ok": True, "compared": "c" * 150}, "roofline": roof, "roofline_other": {"count_A": roof, "ref_flags": roof}, "note": "n" * 700,
            "pmc_kernels": {f"kernel_{i}": {"FETCH_SIZE": 1.0e9, "WRITE_SIZE": 2.0e9} for i in range(40)},
            "secondary": dict({name: leg for name in ("uhgg_deep_focused_sample", "uhgg_deep_focused_snp1pct", "uhgg_focused_sample", "uhgg_default_sample",
                                                      "uhgg_ragged_reference", "uhgg_ragged_deep_focused", "uhgg_packed_reference", "configs1_1g")},
                              configs4_progenomes_1gpu={"k32": leg, "k21": leg, "workload": "q" * 200}, uhgg_error="e" * 500),
            "e2e": {"value": 31.678, "what": "f" * 300, "big": {"sample_1": {"value": 57.98}, "default_sample_2e9": {"value": 47.1}, "what": "g" * 300}},
            "cpu_baseline": {"value": 0.030705, "unit": "M paired-reads/s", "cores": 256, "kind": "port", "sample": "s" * 200, "identical_to_gpu": True,
                             "gpu_same_files": {"total_s": 0.037}, "reference": {"value": 0.0231, "threads": 10, "wall_s": 17.3, "own_clock_s": {"count": 7, "total": 17},
                                                                                "identical_to_gpu": True, "what": "r" * 500}}}


def test_bench_compact_line_stays_short_and_complete():
    """the line the driver parses (tools/benchlib/compact.py): under 4 KB whatever the detail record holds, json round trip, and
    every field VERDICT r3 #1 lists -- round 3's one 23 KB line did not parse and the round went unmeasured"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from benchlib.compact import LIMIT, compact_line
    d = _canned_detail()
    s = json.dumps(compact_line(d))
    assert len(s) < LIMIT == 4096, len(s)
    line = json.loads(s)
    for key in ("metric", "value", "unit", "n_gpus", "world_size", "steps", "warmup", "ms_per_step", "dtype", "config", "phase_ms", "raw_peaks",
                "filtered_peaks", "roofline", "cpu_baseline", "higher_is_better", "scaling", "vs_baseline", "data", "value_found"):
        assert key in line, key
    assert line["vs_baseline"] is None and line["planted_transfers"]["recall"] == 0.0 and line["found"]["recall"] == 1.0
    for key in ("kernel", "bound", "peak", "unit", "launch_ms", "launches_per_step", "traffic", "achieved", "frac", "frac_raw", "frac_model", "model_exceeded",
                "frac_needed", "overfetch", "request_rate"):
        assert key in line["roofline"], key
    assert line["roofline"]["frac"] == round(line["roofline"]["achieved"] / line["roofline"]["peak"], 4)
    assert {"value", "cores", "kind", "identical_to_gpu", "reference"} <= set(line["cpu_baseline"]) and line["cpu_baseline"]["reference"]["threads"] == 10
    assert line["secondary"]["configs1_1g"]["value"] == 134.168 and line["secondary"]["configs4_k21"]["recall"] == 1.0
    assert line["secondary"]["e2e_32m_sample_1"]["value"] == 57.98
    # a record bloated far beyond anything a run produces still yields a parseable line: optional parts are shed, the metric never
    for i in range(60):
        d["secondary"][f"extra_leg_{i}"] = d["secondary"]["configs1_1g"]
    s = json.dumps(compact_line(d))
    assert len(s) < LIMIT and json.loads(s)["value"] == 95.7502 and "roofline" in json.loads(s)
    # the headline alone (the line printed right after the timed steps, before any secondary leg)
    first = {k: v for k, v in _canned_detail().items() if k not in ("secondary", "e2e", "cpu_baseline")}
    line = json.loads(json.dumps(compact_line(first)))
    assert line["value"] == 95.7502 and "value_found" not in line and line["roofline"]["frac_needed"] == 0.0505


def test_bench_needed_bytes_model_and_memory_plan():
    """the needed-bytes model of the roofline (tools/benchlib/roofline.py) on round 3's measured counts, and the per-GPU memory plan
    of BASELINE configs[3]: 125 M pairs + the 156 GB index + tables + key buffers fit 288 GB; a 50 Gbase index does not"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from benchlib.plan import HBM_BYTES, check_fits, memory_plan
    from benchlib.roofline import needed_bytes, rooflines
    stats = {"count_keys": 71_000_000_000, "scan_probes": 16_800_000_000, "vote_l2_probes": 0, "vote_hbm_probes": 1_100_000_000, "vote_revoted_pairs": 10}
    need = needed_bytes(150, 32, 3, 100_000_000, 13_000_000_000, 13000, False, stats, True, "single-first", "queued")
    assert abs(need["count_A"][0] - (100e6 * 2 * 78 + 10 * 71e9 + 12 * 2 * 2 ** 30)) < 1e9      # 24-bit level-1 keys: 3 + 3 + 2 + 2 B per key; 12 chunks of 8 Mi pairs
    assert abs(need["ref_flags"][0] - (16.8e9 * 128 + 12 * (13e9 - 13000 * 31) + 2 * 13e9)) < 1e9
    assert abs(need["vote_kernel"][0] - (100e6 * 2 * 78 + 1.1e9 * 128)) < 1e9
    scan = {"lite": True, "form": "single-first", "frac_slots_at_3": 0.8178, "tiles": 6500000, "tiles_exact": 13001}
    traffic = {"count_A": {"bytes": 955_000_000_000, "bytes_raw": 700_000_000_000}, "ref_flags": {"bytes": 2_300_000_000_000, "bytes_raw": 1_160_000_000_000},
               "vote_kernel": {"bytes": 963_000_000_000, "bytes_raw": 481_000_000_000, "l2_read_requests": 78_400_000_000}}
    roof, dom = rooflines(32, 3, 150, 100_000_000, 13_000_000_000, 13000, False, [285.0, 361.0, 384.0, 317.0], scan, 25119, traffic, "test", stats)
    assert dom == "vote_kernel" and roof[dom]["kernel"] == "vote_kernel_queued"
    for ent in roof.values():
        assert 0 < ent["frac_needed"] <= 1.0 and ent["overfetch"] >= 0.95      # needed bytes never exceed the peak; the counters never show less than needed
    assert roof["vote_kernel"]["model_exceeded"] and roof["vote_kernel"]["overfetch"] > 5
    # what bounds each kernel (round 5): the queued vote by the L2's request rate (78.4 G requests in 384 ms = 204 G/s of the guide's
    # 269.5), phase A by random LDS operations (6 per key), the probe kernel of phase B by the fabric's line rate
    assert (roof["vote_kernel"]["bound"], roof["count_A"]["bound"], roof["ref_flags"]["bound"]) == ("l2_requests", "lds_random", "hbm_lines")
    assert abs(roof["vote_kernel"]["frac_of_bound"] - 78.4 / 0.384 / 269.5) < 0.01 and roof["vote_kernel"]["frac"] < 0.35
    assert abs(roof["count_A"]["frac_of_bound"] - 6 * 71e9 / 0.285 / 1e12 / 3.99) < 0.01
    plan = memory_plan(125_000_000, 13_000_000_000, 13000, world=8)
    assert 250e9 < plan["total"] < 0.95 * HBM_BYTES and abs(plan["reference"] - 156e9) < 1e9 and abs(plan["partition_key_buffers"] - 38.5e9) < 1e9
    check_fits(plan)
    with pytest.raises(SystemExit, match="needs 7"):
        check_fits(memory_plan(25_000_000, 50_000_000_000, 50000), what="50 Gbase as an index")
    check_fits(memory_plan(25_000_000, 50_000_000_000, 50000, packed=True))
    # configs[3] as `bench.py --gpus 8 --ref-form packed` lays it out: 125 M pairs per GPU, the 13 Gbase reference whole on every
    # GPU as 4.9 GB of planes; and configs[4]'s reference sharded over the eight as an index (75 GB each) fits where the whole does not
    p3 = check_fits(memory_plan(125_000_000, 13_000_000_000, 13000, packed=True, world=8))
    assert abs(p3["reference"] - 4.875e9) < 1e7 and p3["exchange_buffers"] == 2 * (1 << 32) // 4 and 100e9 < p3["total"] < 125e9
    # ... and with the slot list of the packed reference next to it (round 5: what `bench.py --gpus 8` lays out by default): 78 GB more
    p3l = check_fits(memory_plan(125_000_000, 13_000_000_000, 13000, packed=True, world=8, slot_list=True))
    assert abs(p3l["slot_list"] - 1.075 * 78e9) < 2e8 and p3l["total"] == p3["total"] + p3l["slot_list"] and p3l["total"] < 205e9
    assert memory_plan(25_000_000, 50_000_000_000, 50000, packed=True, slot_list=True)["slot_list"] == 0     # beyond 2^34 positions: no list
    p4 = check_fits(memory_plan(25_000_000, 50_000_000_000, 50000, world=8, shard_index=True))
    assert abs(p4["reference"] - 75e9) < 1e9
