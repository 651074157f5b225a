"""The real N > 1 path of `extract_ref` with several PROCESSES on the one GPU of the test box: `torch.distributed.run
--nproc-per-node N bin/extract_ref ...`, every rank with its own engine on device 0, the exchanges of localhgt_amd/dist.py staged
through host memory over gloo (RCCL refuses two ranks on one device).  Everything of run(dist=...) executes as on N GPUs:
line counts of 1/N of both FASTQs per rank and their all-gather, each rank parsing its contiguous run of pairs, rank 0 building
the index while the others wait, the packed count-table reduce-scatter, the replicated or reference-sharded scan, the vote
all-reduce -- and the files written must be the reference's goldens (E:1424-1507 is the fork-join this replaces)."""
import os
import re
import shutil
import socket
import subprocess
import sys

import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, argv, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LHGT_EMULATE_THREADS",
                                                             "LHGT_REF_FORM", "LHGT_SHARD_INDEX")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", LHGT_INGEST_CHUNK_BYTES="20000", LHGT_INGEST_TRACE="1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bin", "extract_ref")] + argv
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if res.returncode != 0:
        pytest.fail("world %d run failed:\n%s\n%s" % (world, res.stdout[-1500:], "\n".join(l for l in res.stderr.splitlines() if "Gloo" not in l)[-4000:]), pytrace=False)
    return res


def _check_against_golden(name, fa2, interval, expect_index=True):
    case = cases.CASES[name]
    gold = os.path.join(cases.GOLDEN_DIR, name)
    meta = __import__("json").load(open(os.path.join(gold, "meta.json")))
    assert open(interval).read() == open(os.path.join(gold, "interval.txt")).read()
    assert open(fa2 + ".genome.len.txt").read() == open(os.path.join(gold, "genome.len.txt")).read()
    if expect_index:
        assert cases.sha256_file(f"{fa2}.k{case.k}.h{case.e}.index.dat") == meta["sha256"]["index.dat"]


def _parsed_share(stderr, world):
    """MB of fq1 each rank parsed and the MB of the file, from the LHGT_INGEST_TRACE lines"""
    out = {}
    for m in re.finditer(r"\[lhgt ingest\] part (\d+)/(\d+): \d+ threads, chunks \[(\d+), (\d+)\) of (\d+) = ([\d.]+) MB of ([\d.]+) MB", stderr):
        assert int(m.group(2)) == world
        out[int(m.group(1))] = (float(m.group(6)), float(m.group(7)), int(m.group(4)) - int(m.group(3)), int(m.group(5)))
    return out


MODES = {
    "replicated": {},
    "sharded": {"LHGT_SHARD_INDEX": "1"},
    "packed": {"LHGT_REF_FORM": "packed"},
}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", ["k24_base", "k24_sample_half_cached", "k24_sample_half_fresh", "k24_sample_bases"])
def test_two_ranks_on_one_gpu_write_the_reference_files(case_inputs, tmp_path, name, mode):
    """world 2: interval file, genome.len.txt and index bytes equal the reference's goldens -- with every read kept, with sampling
    by the global read ordinal (a cached index and one built in the run by rank 0: quirk Q3's stream position), with --sample > 1
    (the base count of cal_sam_ratio from the ranks' line-length sums)"""
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    interval = str(tmp_path / "S.interval.txt")
    argv = cases.extract_ref_argv(case, f1, f2, fa2, interval)
    if case.preexisting_index:                      # the golden was made with the index already in place
        from localhgt_amd import extract_ref
        extract_ref.run(extract_ref.parse_argv(cases.extract_ref_argv(case, f1, f2, fa2, str(tmp_path / "first.txt"))), log=lambda *a: None)
    res = _launch(2, argv, MODES[mode])
    _check_against_golden(name, fa2, interval, expect_index=(mode != "packed" or case.preexisting_index))
    share = _parsed_share(res.stderr, 2)
    assert set(share) == {0, 1}, res.stderr[-2000:]
    total_mb = share[0][1]
    for r in (0, 1):                                # each rank parsed about half of fq1, not all of it
        assert 0.3 * total_mb < share[r][0] < 0.7 * total_mb, share
    assert share[0][2] + share[1][2] == share[0][3]
    if mode == "sharded":
        assert "lines of bytes" in res.stderr or "lines of chunks" in res.stderr     # every rank counted its share of the lines (either planner)


@pytest.mark.parametrize("name,world,mode", [
    ("k24_t10_sample_bases", 2, "replicated"),      # the CLI's default -t 10: thread emulation under N ranks
    ("k24_t4", 2, "sharded"),                       # ... with the contig groups of split_ref cut across the ranks' shards
    ("k24_t8_sample_half", 3, "sharded"),
    ("k24_t3_fq2_longer", 2, "packed"),
    ("k24_fq2_stray2", 2, "replicated"),            # fq2 re-synchronised on fq1's first read ID (E:368-402), foreign records on rank 0
    ("k24_fq2_surplus", 3, "replicated"),           # fq2's surplus records on the last rank
    ("k24_fq2_short_nonl", 2, "sharded"),           # fq2 runs out: the stale line std::getline leaves behind
    ("k24_nrun_lower", 4, "sharded"),
    ("k32_base", 2, "replicated"),                  # 2^32-slot tables: 1 GiB of packed counts through the host-staged reduce-scatter
])
def test_more_ranks_and_the_reference_corner_cases(case_inputs, tmp_path, name, world, mode):
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    interval = str(tmp_path / "S.interval.txt")
    res = _launch(world, cases.extract_ref_argv(case, f1, f2, fa2, interval), MODES[mode])
    _check_against_golden(name, fa2, interval, expect_index=(mode != "packed"))
    if case.threads > 1:
        assert f"reproducing the reference's -t {case.threads}" in res.stdout
        assert "warning" not in res.stdout
    share = _parsed_share(res.stderr, world)
    assert len(share) == world and sum(s[2] for s in share.values()) == share[0][3]


@pytest.mark.parametrize("idx", range(int(os.environ.get("LHGT_WORLD_FUZZ_FIRST", "0")),
                                      int(os.environ.get("LHGT_WORLD_FUZZ_FIRST", "0")) + int(os.environ.get("LHGT_WORLD_FUZZ_CASES", "6"))))
def test_random_case_with_several_ranks_matches_oracle(oracle, tmp_path, idx):
    """the random whole runs of tests/test_gpu_fuzz.py through 2-4 PROCESSES (replicated / reference-sharded / packed by turns, every
    third case as `-t N`): the files must be the CPU restatement's -- its `-t N` form where the product emulated the thread chunks
    (its log says so), the `-t 1` form otherwise.  LHGT_WORLD_FUZZ_CASES / LHGT_WORLD_FUZZ_FIRST lengthen / move the run."""
    import numpy as np
    from test_gpu_fuzz import _make_case
    g, c = tmp_path / "gpu", tmp_path / "cpu"
    g.mkdir()
    k, e, seed, sample, hit, match, max_peak = _make_case(5000 + idx, str(g))
    shutil.copytree(g, c, dirs_exist_ok=True)
    world = 2 + idx % 3
    mode = list(MODES)[idx % 3]
    threads = 2 + idx % 7 if idx % 3 == 2 else 1
    argv = [str(g / "s.1.fq"), str(g / "s.2.fq"), str(g / "ref.fa"), str(g / "i.txt"), repr(hit), repr(match), str(threads), str(k),
            str(max_peak), str(e), str(seed), repr(sample)]
    env = {kk: v for kk, v in os.environ.items() if kk not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LHGT_EMULATE_THREADS",
                                                                "LHGT_REF_FORM", "LHGT_SHARD_INDEX")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", LHGT_INGEST_CHUNK_BYTES="20000", **MODES[mode])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bin", "extract_ref")] + argv
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    o = (str(c / "s.1.fq"), str(c / "s.2.fq"), str(c / "ref.fa"), str(c / "i.txt"), float(np.float32(hit)), float(np.float32(match)))
    emulated = threads > 1 and "reproducing the reference's -t" in res.stdout and "giving the -t 1 result" not in res.stdout
    if emulated:
        rc, orep = oracle.run_threads(*o, threads, k, max_peak, e, seed, sample)
    else:
        rc, orep = oracle.run(*o, 1, k, max_peak, e, seed, sample)
    if res.returncode != 0:                       # refused on the GPU side: the restatement of the same -t must refuse too
        rc_t = oracle.run_threads(*o, threads, k, max_peak, e, seed, sample)[0] if threads > 1 else rc
        assert rc != 0 or rc_t != 0, (res.returncode, rc, rc_t, mode, world, threads, k, e, sample, res.stderr[-1500:])
        return
    assert rc == 0, (rc, mode, world, threads, k, e, sample, res.stdout[-800:])
    names = ("i.txt", "ref.fa.genome.len.txt") + (() if mode == "packed" else (f"ref.fa.k{k}.h{e}.index.dat",))
    for name in names:
        assert open(g / name, "rb").read() == open(c / name, "rb").read(), (name, mode, world, threads, k, e, seed, sample, hit, match)


@pytest.mark.parametrize("mode", ["replicated", "packed"])
def test_two_ranks_run_a_batch(case_inputs, tmp_path, mode):
    """`extract_ref --batch MANIFEST` (round 6) under two ranks: three samples of one reference in one process group -- every rank keeps
    the reference resident between the samples, the exchanges of every sample run in turn -- and every sample's files are the goldens
    (the first sample builds the index with every read kept; the sampled one is the golden made on an index in place)"""
    names = ["k24_base", "k24_sample_half_cached", "k24_t4"]
    fa2 = str(tmp_path / "ref.fa")
    lines, outs = [], []
    for i, name in enumerate(names):
        case = cases.CASES[name]
        fa, f1, f2, meta = case_inputs(name)
        if i == 0:
            shutil.copy(fa, fa2)
        interval = str(tmp_path / f"s{i}.interval.txt")
        lines.append(" ".join(cases.extract_ref_argv(case, f1, f2, fa2, interval)))
        outs.append((name, interval))
    if mode == "packed":       # the coder header the cached golden was made with: an index in place before the batch
        from localhgt_amd import extract_ref
        a0 = extract_ref.parse_argv(lines[0].split())
        extract_ref.run(a0, log=lambda *a: None)
        os.remove(a0.interval)
    manifest = tmp_path / "batch.txt"
    manifest.write_text("\n".join(lines) + "\n")
    res = _launch(2, ["--batch", str(manifest)], MODES[mode])
    assert res.stdout.count("reference: resident from the previous sample") >= 2, res.stdout[-3000:]     # samples 2 and 3 (on every rank whose stdout the launcher passes on)
    for name, interval in outs:
        _check_against_golden(name, fa2, interval, expect_index=True)
