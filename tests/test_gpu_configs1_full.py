"""BASELINE configs[1] at FULL size against the REAL reference binary (round 6: collected, so the driver's `-m gpu` run sees it).

tests/golden/configs1_full/ holds what /root/reference/src/extract_ref_normal_peak.cpp, compiled unmodified (oracle/build_ref.sh),
wrote in the build container for 1000 x 1 Mbp + 10 M pairs at k = 32, e = 3 -- its interval files for `-t 1` and for `-t 10` (threads
in creation order, oracle/seq_threads.c) and the sha256 of its 12 GB index and of genome.len.txt -- from the files of
tests/synth_cpu.c, the host twin of the device generator.  Here: the device generator writes the same files (inputs.sha256 proves
it), the product runs them as `-t 1` (index built in-run) and `-t 10` (the CLI's default: thread emulation, index cached), and
every output is compared byte for byte.  The hashing of 19 GB of files runs on host threads next to the GPU runs."""
import hashlib
import os
import shutil
import tempfile
import threading

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "configs1_full")
NC, PAIRS, K, E = 1000, 10_000_000, 32, 3


def _sha(path, out, key, zero_at=None):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        if zero_at:
            head = bytearray(f.read(zero_at[1]))
            head[zero_at[0]:zero_at[1]] = bytes(zero_at[1] - zero_at[0])
            h.update(bytes(head))
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    out[key] = h.hexdigest()


def test_configs1_full_size_equals_the_reference_binary():
    import bench
    from localhgt_amd import extract_ref
    want_in = dict(reversed(line.split(None, 1)) for line in open(os.path.join(GOLD, "inputs.sha256")).read().splitlines())
    want_out = [line.split()[0] for line in open(os.path.join(GOLD, "outputs.sha256")).read().splitlines()]
    tmp = tempfile.mkdtemp(prefix="lhgt_c1full_", dir=tempfile.gettempdir())
    try:
        if shutil.disk_usage(tmp).free < 40e9:
            pytest.skip("needs 40 GB of scratch space for the files of configs[1]")
        fa, f1, f2 = bench.synth_files(tmp, K, E, NC, 1_000_000, PAIRS, 0)
        sums, threads = {}, []
        for p in (fa, f1, f2):
            threads.append(threading.Thread(target=_sha, args=(p, sums, os.path.basename(p))))
            threads[-1].start()
        got = {}
        for t in (1, 10):
            out = os.path.join(tmp, f"gpu_t{t}.txt")
            a = extract_ref.Args(f1, f2, fa, out, 0.1, 0.08, t, K, 300_000_000, E, 1, 1.0)
            rep = extract_ref.run(a, log=lambda *x: None)
            assert rep["emulated_threads"] == t and rep["pairs_kept"] <= PAIRS and rep["index_built"] == (t == 1)
            got[t] = open(out, "rb").read()
            if t == 1:      # the index and genome.len.txt are final: hash them while the -t 10 run goes on
                idx = f"{fa}.k{K}.h{E}.index.dat"
                threads.append(threading.Thread(target=_sha, args=(idx, sums, "index", (1198, 1200))))   # bytes 1198-1199: whatever lies behind the reference's coder array
                threads[-1].start()
                threads.append(threading.Thread(target=_sha, args=(fa + ".genome.len.txt", sums, "genome.len")))
                threads[-1].start()
        for th in threads:
            th.join()
        for name in ("ref.fa", "s.1.fq", "s.2.fq"):
            assert sums[name] == want_in[name].strip(), f"the device generator's {name} is not the file the golden was made from (tests/synth_cpu.c)"
        for t in (1, 10):
            assert got[t] == open(os.path.join(GOLD, f"interval_t{t}.txt"), "rb").read(), f"-t {t}: interval file differs from the reference binary's"
        assert got[1] != got[10] and got[10].count(b"\n") > got[1].count(b"\n")      # ten sentinel lines instead of one (E:520-543)
        assert [sums["genome.len"], sums["index"]] == want_out, "genome.len.txt / index file differ from the reference binary's"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
