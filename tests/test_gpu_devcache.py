"""The process-wide cache of large device blocks (localhgt_amd/csrc/cabi.hip: dev_free / dev_alloc_raw) under two contexts on one
GPU: a block one context parks while its kernels are still queued -- key buffers and read batches that regrow in
mid-stream -- may be handed to the other context only once the device has drained what was queued before the park.  Two host
threads, each with a context of its own, grow their batches step by step (every step frees and re-allocates blocks of 64 MiB and
more, of size classes the two contexts share) and count; every table must equal the one a lone context computes."""
import os
import threading

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, E, NC, CL = 26, 3, 40, 100_000
STEPS = [60_000, 150_000, 90_000, 400_000, 250_000, 700_000]


def _table(eng, seed, n):
    eng.pairs_clear()
    eng.counts_clear()
    eng.synth_pairs(1, seed, NC, CL, 0, n)
    eng.set_count_mode(1)                  # the radix partition: key buffers sized by the batch
    eng.count_kmers()
    return eng.digest(eng.DIGEST_COUNTS)


def test_two_contexts_share_the_block_cache_while_regrowing():
    from localhgt_amd.engine import Engine, pool_trim
    pool_trim()
    with Engine(K, E) as lone:
        lone.rng_seed(1)
        lone.coder_generate()
        lone.synth_reference(1, NC, CL)
        want = {(seed, n): _table(lone, seed, n) for seed in (2, 3) for n in STEPS}
    got, errs = {}, []

    def worker(seed):
        try:
            for rnd in range(2):
                with Engine(K, E) as eng:      # closing parks the context's blocks for the other thread's next regrowth
                    eng.rng_seed(1)
                    eng.coder_generate()
                    eng.synth_reference(1, NC, CL)
                    for n in (STEPS if rnd == 0 else STEPS[::-1]):
                        got[(seed, n, rnd)] = _table(eng, seed, n)
        except Exception as ex:               # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=worker, args=(seed,)) for seed in (2, 3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for (seed, n, rnd), d in got.items():
        assert d == want[(seed, n)], (seed, n, rnd)
    pool_trim()


def test_an_allocation_out_of_memory_drops_the_idle_slot_list(case_inputs, tmp_path):
    """the slot list is optional (78-130 GB on a catalogue) and a later, larger sample must not fail where it would have fit
    without it.  With LHGT_TEST_FAIL_ALLOC the first attempt of a larger allocation fails as on a full device while the context holds an
    idle list: the list goes, the allocation succeeds, the next scan takes the position-ordered kernel -- same interval file.
    (A subprocess: the hook is read once per process.)"""
    import subprocess
    import sys
    import cases
    case = cases.CASES["k24_base"]
    fa, f1, f2, meta = case_inputs("k24_base")
    script = f"""
import sys
sys.path.insert(0, {ROOT!r})
from localhgt_amd.engine import Engine
eng = Engine(24, 3)
eng.rng_seed(1); eng.coder_generate(); eng.set_reference_form(True)
eng.reference_load_fasta({fa!r})
eng.slot_list(2)
eng.sampling_init(100.0)
eng.pairs_load_fastq({f1!r}, {f2!r}, 100.0)
eng.count_kmers()
n = eng.ref_scan(0.1, 0.08, 1000000)
a_entries = eng.slot_list()["entries"]                       # (the probe kernels ran on the list; the first allocation behind them -- the peak registry's -- already dropped it)
eng.vote()
a = (n, eng.scan_info()["form"], a_entries, eng.digest(eng.DIGEST_VOTES), eng.digest(eng.DIGEST_PEAK_KMER))
eng.pairs_clear()                                            # the next sample: its batches are new allocations
eng.pairs_load_fastq({f1!r}, {f2!r}, 100.0)
b_entries = eng.slot_list()["entries"]
eng.counts_clear(); eng.count_kmers()
n2 = eng.ref_scan(0.1, 0.08, 1000000); eng.vote()
b = (n2, eng.scan_info()["form"], eng.slot_list()["entries"], eng.digest(eng.DIGEST_VOTES), eng.digest(eng.DIGEST_PEAK_KMER))
print(a); print(b_entries); print(b)
assert a[1] == "slot-first" and a[2] == 0, a
assert b_entries == 0, "the list should have been dropped by the failed allocation"
assert b[1] == "trio-first" and b[2] == 0, b        # ... and is not built again for this reference
assert (a[0], a[3], a[4]) == (b[0], b[3], b[4]), (a, b)
"""
    env = dict(os.environ, LHGT_TEST_FAIL_ALLOC="4096", LHGT_TRACE="1")
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert "the slot list was dropped" in res.stderr


def test_slot_list_from_a_sampled_histogram_and_its_fallback(tmp_path):
    """round 6 (late): the slot list's regions come from the histogram of every 8th tile + slack; a bucket that still runs over -- here
    9 000 consecutive A's (one k-mer, one slot) inside tiles the sample does not look at -- makes the build start over with the exact
    histogram.  Both builds, and LHGT_SLOT_LIST_SAMPLE=0, give the flags, loci, peak_kmer and votes of the position-ordered kernel.
    (Subprocesses: the variable is read once per process.)"""
    import subprocess
    import sys
    import numpy as np
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_contigs, clen = 40, 100_000
    contigs = [acgt[rng.integers(0, 4, clen)].copy() for _ in range(n_contigs)]
    fa_plain, fa_rep = str(tmp_path / "plain.fa"), str(tmp_path / "rep.fa")
    for path, rep in ((fa_plain, False), (fa_rep, True)):
        with open(path, "wb") as f:
            for i, c in enumerate(contigs):
                c = c.copy()
                if rep and i == 0:
                    c[2500:11500] = ord("A")              # tiles 1 .. 5 of contig 0 (2000 positions each): not among every 8th tile
                f.write(b">c%d\n" % i + c.tobytes() + b"\n")
    reads = []
    for _ in range(20000):
        c = contigs[int(rng.integers(0, 8))]
        p = int(rng.integers(0, clen - 150))
        reads.append(c[p:p + 150].tobytes())
    f1, f2 = str(tmp_path / "s.1.fq"), str(tmp_path / "s.2.fq")
    for path, tag in ((f1, b"/1"), (f2, b"/2")):
        with open(path, "wb") as f:
            for i, r in enumerate(reads):
                f.write(b"@r%d%s\n" % (i, tag) + r + b"\n+\n" + b"I" * 150 + b"\n")

    def run(fa, sample_env):
        script = f"""
import sys
sys.path.insert(0, {ROOT!r})
from localhgt_amd.engine import Engine
eng = Engine(24, 3)
eng.rng_seed(1); eng.coder_generate(); eng.set_reference_form(True)
eng.reference_load_fasta({fa!r})
eng.sampling_init(100.0)
eng.pairs_load_fastq({f1!r}, {f2!r}, 100.0)
eng.count_kmers()
out = []
for dbg in (16384, 1 << 24):                                 # the trio-first kernel, then the list (built now)
    eng.set_debug(dbg)
    n = eng.ref_scan(0.1, 0.08, 1000000)
    eng.vote()
    out.append((n, eng.scan_info()["form"], eng.digest(eng.DIGEST_LOCI), eng.digest(eng.DIGEST_PEAK_KMER), eng.digest(eng.DIGEST_FLAGS, 0b1111100), eng.digest(eng.DIGEST_VOTES)))
print(out, eng.slot_list())
assert out[0][1] == "trio-first" and out[1][1] == "slot-first", out
assert out[0][0] == out[1][0] and out[0][2:] == out[1][2:], out
assert eng.slot_list()["entries"] > 0
"""
        env = dict(os.environ, LHGT_TRACE="1")
        if sample_env is not None:
            env["LHGT_SLOT_LIST_SAMPLE"] = sample_env
        res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
        return res.stderr

    err = run(fa_plain, None)
    assert "regions from a sampled histogram" in err and "built again" not in err, err[-2000:]
    err = run(fa_rep, None)
    assert "built again from the exact one" in err and "regions from the exact histogram" in err, err[-2000:]
    err = run(fa_rep, "0")
    assert "regions from the exact histogram" in err and "built again" not in err, err[-2000:]
