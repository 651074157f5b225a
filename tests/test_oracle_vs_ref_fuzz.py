"""Extra pinning of the CPU restatement: the same random-case generator as tests/test_gpu_fuzz.py, but the oracle is
compared with the REAL reference binary (oracle/_ref/extract_ref_z, built from /root/reference by oracle/build_ref.sh).
Skipped where that binary is absent.  Cases the reference handles with undefined behaviour are left out: max_peak
overflow (it writes past its arrays, E:272-274) and contigs shorter than half a window with a zero threshold
(zero-sized interval array, E:565)."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from test_gpu_fuzz import _apply_variant, _make_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "extract_ref_z")

pytestmark = pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/extract_ref_z not built (needs /root/reference)")


N_CASES = int(os.environ.get("LHGT_REF_FUZZ_CASES", "16"))     # LHGT_REF_FUZZ_CASES=300 for a soak of the restatement against the binary


@pytest.mark.parametrize("idx", range(N_CASES))
def test_oracle_equals_reference_binary(oracle, tmp_path, idx):
    r, c = tmp_path / "ref", tmp_path / "cpu"
    r.mkdir()
    k, e, seed, sample, hit, match, max_peak = _make_case(idx, str(r), k_max=26)
    if max_peak < 1000:
        max_peak = 100000
    if hit == 0.0 or match == 0.0:
        hit, match = max(hit, 0.05), max(match, 0.02)
    _apply_variant(idx, r)                      # every seventh case in "\r\n" line ends, every eleventh with blank lines in the FASTA
    shutil.copytree(r, c, dirs_exist_ok=True)
    runs = 2 if idx % 3 == 0 else 1
    for _ in range(runs):
        res = subprocess.run([REF_BIN, "s.1.fq", "s.2.fq", "ref.fa", "i.txt", repr(hit), repr(match), "1", str(k), str(max_peak), str(e),
                              str(seed), repr(sample)], cwd=r, capture_output=True, text=True, timeout=300)
        rc, orep = oracle.run(str(c / "s.1.fq"), str(c / "s.2.fq"), str(c / "ref.fa"), str(c / "i.txt"), float(np.float32(hit)),
                              float(np.float32(match)), 1, k, max_peak, e, seed, sample)
        if idx % 7 == 6 and rc in (-4, -6):     # a SAMPLED line of more than 500 characters (500 bases + "\r"): the reference writes past
            return                              # its stack buffers (by one entry it survives, by more it aborts), the restatement refuses
        assert res.returncode == 0, res.stderr[-500:]
        assert rc == 0
    raw = re.findall(r"No\. of raw BKPs: (\d+)", res.stdout)   # absent when no contig is longer than k (no scan thread starts)
    assert (int(raw[-1]) if raw else 0) == orep.n_peaks
    assert open(r / "i.txt").read() == open(c / "i.txt").read(), (k, e, seed, sample, hit, match)
    assert open(r / "ref.fa.genome.len.txt").read() == open(c / "ref.fa.genome.len.txt").read()
    a, b = open(r / f"ref.fa.k{k}.h{e}.index.dat", "rb").read(), open(c / f"ref.fa.k{k}.h{e}.index.dat", "rb").read()
    assert len(a) == len(b) and a[:1198] == b[:1198] and a[1200:] == b[1200:]   # bytes 1198-1199: past-the-array read in the reference


SHIM = os.path.join(ROOT, "oracle", "_ref", "libseqthreads.so")


@pytest.mark.skipif(not os.path.exists(SHIM), reason="oracle/_ref/libseqthreads.so not built")
@pytest.mark.parametrize("idx", range(N_CASES))
def test_oracle_thread_emulation_equals_reference_binary(oracle, tmp_path, idx):
    """-t N: the restatement's thread chunks, per-chunk sampling ordinals, contig groups, id ranges and
    per-thread sentinel lines against the reference run with its threads in creation order (oracle/seq_threads.c)"""
    r, c = tmp_path / "ref", tmp_path / "cpu"
    r.mkdir()
    k, e, seed, sample, hit, match, max_peak = _make_case(100 + idx, str(r), k_max=24)
    threads = 2 + idx % 9
    max_peak = 100000 * threads
    if hit == 0.0 or match == 0.0:
        hit, match = max(hit, 0.05), max(match, 0.02)
    shutil.copytree(r, c, dirs_exist_ok=True)
    rc, orep = oracle.run_threads(str(c / "s.1.fq"), str(c / "s.2.fq"), str(c / "ref.fa"), str(c / "i.txt"), float(np.float32(hit)),
                                  float(np.float32(match)), threads, k, max_peak, e, seed, sample)
    if rc in (-4, -6):
        pytest.skip("a thread chunk starts within 1000 bytes of EOF / fq2 cannot be re-synchronised: the reference reads garbage there")
    assert rc == 0
    res = subprocess.run([REF_BIN, "s.1.fq", "s.2.fq", "ref.fa", "i.txt", repr(hit), repr(match), str(threads), str(k), str(max_peak), str(e),
                          str(seed), repr(sample)], cwd=r, capture_output=True, text=True, timeout=300, env=dict(os.environ, LD_PRELOAD=SHIM))
    assert res.returncode == 0, res.stderr[-500:]
    assert open(r / "i.txt").read() == open(c / "i.txt").read(), (threads, k, e, seed, sample, hit, match)
    mates = [int(x) for x in re.findall(r">>> Thread: final read .* read num: (\d+)", res.stdout)]
    assert sum(mates[:threads]) == orep.pairs_counted and sum(mates[threads:]) == int(orep.t_count)   # reads each chunk kept, mate 1 / mate 2


import cases


@pytest.mark.parametrize("name,text", cases.odd_fastas(), ids=[n for n, _ in cases.odd_fastas()])
def test_oracle_reads_odd_fastas_like_the_reference_binary(oracle, tmp_path, name, text):
    """read_ref's line semantics (E:761-880) on FASTA files with unusual structure: the restatement against the reference binary,
    index bytes, genome.len.txt and interval file"""
    k, e = 16, 3
    rng = np.random.default_rng(len(text))
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=100).tolist()) for _ in range(4)]
    for d in ("ref", "cpu"):
        (tmp_path / d).mkdir()
        (tmp_path / d / "ref.fa").write_bytes(text)
        for m in (1, 2):
            with open(tmp_path / d / f"s.{m}.fq", "wb") as f:
                for i, r in enumerate(reads):
                    f.write(b"@q%d/%d\n" % (i, m) + r + b"\n+\n" + b"I" * len(r) + b"\n")
    r, c = tmp_path / "ref", tmp_path / "cpu"
    res = subprocess.run([REF_BIN, "s.1.fq", "s.2.fq", "ref.fa", "i.txt", "0.1", "0.08", "1", str(k), "100000", str(e), "1", "1"],
                         cwd=r, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-500:]
    rc, _ = oracle.run(str(c / "s.1.fq"), str(c / "s.2.fq"), str(c / "ref.fa"), str(c / "i.txt"), float(np.float32(0.1)), float(np.float32(0.08)),
                       1, k, 100000, e, 1, 1.0)
    assert rc == 0
    assert open(r / "ref.fa.genome.len.txt").read() == open(c / "ref.fa.genome.len.txt").read()
    a, b = open(r / f"ref.fa.k{k}.h{e}.index.dat", "rb").read(), open(c / f"ref.fa.k{k}.h{e}.index.dat", "rb").read()
    assert len(a) == len(b) and a[:1198] == b[:1198] and a[1200:] == b[1200:]
    assert open(r / "i.txt").read() == open(c / "i.txt").read()
