#!/usr/bin/env python3
"""where tests/test_gpu_fullsize_uhgg.py spends its time (round 6: the GPU suite's budget): the steps of its two longest tests, timed one by one"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from localhgt_amd.engine import Engine
NC, CL, K, E = 13000, 1_000_000, 32, 3
T0 = [time.time()]
def lap(tag):
    t = time.time(); print(f"{t - T0[0]:8.2f} s  {tag}", flush=True); T0[0] = t
eng = Engine(K, E); eng.rng_seed(1); eng.coder_generate(); lap("engine")
eng.synth_reference(1, NC, CL); eng.synchronize(); lap("synth_reference index form")
eng.synth_options(0, 20, 1000); eng.synth_pairs(1, 2, NC, CL, 0, 25_000_000); eng.synth_options(0, 20, 0); eng.synchronize(); lap("synth_pairs 25 M")
eng.counts_clear(); eng.count_kmers(); lap("count")
for dbg in (8192, 0, 4096, 16384):
    eng.set_debug(dbg); n = eng.ref_scan(0.1, 0.08, 300_000_000); lap(f"scan dbg {dbg}: {n} peaks, {eng.scan_info()['form']}, kernel {eng.phase_ms(1):.0f} ms")
    eng.digest(eng.DIGEST_LOCI); eng.digest(eng.DIGEST_PEAK_KMER); eng.digest(eng.DIGEST_FLAGS, 0b1111100); lap("  3 digests")
eng.set_debug(0); eng.ref_scan(0.1, 0.08, 300_000_000); eng.vote(); lap(f"scan+vote, vote kernel {eng.phase_ms(2):.0f} ms")
eng.set_reference_form(True); lap("set_reference_form(True): index dropped")
eng.synth_reference(1, NC, CL); eng.synchronize(); lap("synth_reference packed")
eng.slot_list(0)
for dbg in (8192, 0, 4096, 16384):
    eng.set_debug(dbg); n = eng.ref_scan(0.1, 0.08, 300_000_000); lap(f"packed scan dbg {dbg}: {eng.scan_info()['form']}, kernel {eng.phase_ms(1):.0f} ms")
eng.slot_list(1); eng.set_debug(1 << 24); eng.ref_scan(0.1, 0.08, 300_000_000); lap(f"slot-first incl. list build, kernel {eng.phase_ms(1):.0f} ms")
eng.set_debug(4096 | (1 << 24)); eng.ref_scan(0.1, 0.08, 300_000_000); lap(f"slot-single incl. list swap, kernel {eng.phase_ms(1):.0f} ms")
eng.set_debug(0); eng.slot_list(0); lap("list dropped")
import bigaddr, oracle_api
from conftest import build_oracle
orc = oracle_api.Oracle(build_oracle())
import tempfile
with tempfile.TemporaryDirectory() as tmp:
    bigaddr.check_against_oracle(eng, orc, tmp, NC, CL, K, E, bigaddr.boundary_contigs(NC, CL, K, E, True)); lap("bigaddr.check_against_oracle (packed)")
eng.close(); lap("close")
