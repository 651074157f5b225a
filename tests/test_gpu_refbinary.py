"""The product at k = 32 beyond the size of the committed goldens -- 20 x 1 Mbp, 400 000 pairs from files, 4 GiB count table,
16 GiB peak_kmer -- against (i) the CPU restatement (oracle.run, always) and (ii) the REAL reference binary
(oracle/_ref/extract_ref_z, built by oracle/build_ref.sh from /root/reference with the zero-new[] shim; skipped where it is absent):
`-t 1`, `-t 10` with the reference's threads in creation order (oracle/_ref/libseqthreads.so) against the product's thread
emulation, and the packed reference form.  Index bytes, genome.len.txt and interval file must be identical.
The reference needs about two minutes per run on the GPU box's host: both runs start in the background at the start of the session
(tests/conftest.py: refbin_big) and are joined by the test that needs them.  Plus: phase A at k = 32 with two partition chunks and overflowing bucket regions
(poly-A) against the oracle's whole 2^32-slot table."""
import os
import shutil
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "extract_ref_z")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libseqthreads.so")


from conftest import REFBIN_K as K, REFBIN_E as E, REFBIN_PAIRS as PAIRS, refbin_copy_inputs as _copy_inputs   # noqa: E402


@pytest.fixture(scope="module")
def big(refbin_big):
    """inputs and the two background runs of the reference binary: made at the START of a -m gpu session (tests/conftest.py), so the
    reference's two minutes per run pass while the earlier test files execute"""
    return refbin_big


def _product(big, tag, threads, ref_form="index"):
    from localhgt_amd import extract_ref
    d = _copy_inputs(big["src"], os.path.join(big["base"], tag))
    a = extract_ref.Args(os.path.join(d, "s.1.fq"), os.path.join(d, "s.2.fq"), os.path.join(d, "ref.fa"), os.path.join(d, "i.txt"), 0.1, 0.08,
                         threads, K, 3_000_000, E, 1, 1.0)
    rep = extract_ref.run(a, log=lambda *x: None, ref_form=ref_form)
    return d, rep


def _same_files(a, b, index=True):
    for name in ("i.txt", "ref.fa.genome.len.txt"):
        assert open(os.path.join(a, name), "rb").read() == open(os.path.join(b, name), "rb").read(), name
    if index:
        x, y = (np.fromfile(os.path.join(d, f"ref.fa.k{K}.h{E}.index.dat"), dtype=np.uint8) for d in (a, b))
        assert x.size == y.size and (x[:1198] == y[:1198]).all() and (x[1200:] == y[1200:]).all()   # bytes 1198-1199: the reference reads past its coder array


def test_product_equals_the_cpu_restatement_at_k32_from_files(big, oracle):
    d_cpu = _copy_inputs(big["src"], os.path.join(big["base"], "cpu"))
    rc, rep = oracle.run(os.path.join(d_cpu, "s.1.fq"), os.path.join(d_cpu, "s.2.fq"), os.path.join(d_cpu, "ref.fa"), os.path.join(d_cpu, "i.txt"),
                         0.1, 0.08, os.cpu_count() or 1, K, 3_000_000, E, 1, 1.0)
    assert rc == 0 and rep.n_peaks > 1000
    d_gpu, rep_g = _product(big, "gpu_t1", 1)
    assert (rep_g["n_peaks"], rep_g["n_filtered"], rep_g["pairs_kept"]) == (rep.n_peaks, rep.n_filtered, PAIRS)
    _same_files(d_cpu, d_gpu)
    assert sum(1 for _ in open(os.path.join(d_gpu, "i.txt"))) > 5
    d_pk, rep_p = _product(big, "gpu_t1_packed", 1, ref_form="packed")
    assert not os.path.exists(os.path.join(d_pk, f"ref.fa.k{K}.h{E}.index.dat"))
    _same_files(d_cpu, d_pk, index=False)
    big["cpu_dir"] = d_cpu


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/extract_ref_z is not built (oracle/build_ref.sh needs /root/reference)")
def test_product_equals_the_reference_binary_at_k32(big):
    """-t 1; -t 10 (threads in creation order) against the thread emulation, which bin/extract_ref applies by default; packed form"""
    outs = {}
    for t, (d, p, t0) in big["procs"].items():
        out, _ = p.communicate(timeout=1500)
        assert p.returncode == 0, out[-2000:]
        outs[t] = out
        print(f"reference binary -t {t}: {time.time() - t0:.0f} s since start; " + " | ".join(l for l in out.splitlines() if "raw BKPs" in l)[:200])
    d1, rep1 = _product(big, "vs_ref_t1", 1)
    _same_files(big["procs"][1][0], d1)
    d10, rep10 = _product(big, "vs_ref_t10", 10)
    assert rep10["emulated_threads"] == 10
    _same_files(big["procs"][10][0], d10)
    assert open(os.path.join(d10, "i.txt")).read().count("1\t1\t1\n") >= 2            # a sentinel line per thread without voted peaks
    assert open(os.path.join(d10, "i.txt")).read() != open(os.path.join(d1, "i.txt")).read()
    dp, repp = _product(big, "vs_ref_t10_packed", 10, ref_form="packed")
    _same_files(big["procs"][10][0], dp, index=False)
    raw = int([l for l in outs[1].splitlines() if "raw BKPs" in l][-1].split("raw BKPs:")[1].split()[0])
    assert raw == rep1["n_peaks"]


def test_phase_a_at_k32_two_chunks_and_overflowing_regions_against_the_oracle_table(oracle, tmp_path):
    """two resident batches (= two partition chunks), each with a poly-A / poly-AC spike whose keys overflow their bucket regions
    (k_count_part.hip: the direct CAS path behind the partition), at k = 32: the WHOLE 2^32-slot table equals the oracle's"""
    from localhgt_amd.engine import Engine
    rng = np.random.default_rng(32)
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)

    def reads(n):
        return [acgt[rng.choice(5, size=int(rng.integers(100, 260)), p=[.248, .248, .248, .248, .008])].tobytes() for _ in range(n)]

    halves = []
    for h in range(2):
        r1 = reads(20000) + [b"A" * 200, b"ACAC" * 50, b"ACGT" * 40] * 300
        r2 = reads(20000) + [b"T" * 200, b"GTGT" * 50, b"TTTTTTTTTTG" * 18] * 300
        halves.append((r1, r2))
    fqs = []
    for m in (0, 1):
        path = str(tmp_path / f"s.{m + 1}.fq")
        with open(path, "wb") as f:
            n = 0
            for h in halves:
                for r in h[m]:
                    f.write(b"@r%d/%d\n" % (n, m + 1) + r + b"\n+\n" + b"I" * len(r) + b"\n")
                    n += 1
        fqs.append(path)
    oracle.srand(11)
    cc = oracle.random_coder(K, E)
    table = np.zeros(1 << K, dtype=np.uint8)
    big_limit = 1 << 40
    for p in fqs:
        assert oracle.count(p, big_limit, K, E, cc, 100.0, None, table, threads=os.cpu_count() or 1) == 2 * 20900
    with Engine(K, E) as eng:
        eng.rng_seed(11)
        eng.coder_generate()
        assert (eng.coder_get() == cc).all()
        eng.set_count_mode(1)
        for r1, r2 in halves:                       # one batch per call
            s1 = np.frombuffer(b"".join(r1), dtype=np.uint8)
            s2 = np.frombuffer(b"".join(r2), dtype=np.uint8)
            o1 = np.cumsum([0] + [len(r) for r in r1]).astype(np.uint64)
            o2 = np.cumsum([0] + [len(r) for r in r2]).astype(np.uint64)
            eng.pairs_append(s1, o1, s2, o2)
        eng.count_kmers()
        hist = eng.counts_histogram()
        assert (hist == np.bincount(table, minlength=4).astype(np.uint64)).all()
        step = 1 << 28
        for first in range(0, 1 << K, step):        # the whole table, a quarter GiB at a time
            assert (eng.counts_export(first, step) == table[first:first + step]).all(), first
