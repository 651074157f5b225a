#!/usr/bin/env python3
"""phase C on the ragged catalogue under the half-of-everything sample (100 M pairs), stage by stage (LHGT_DEBUG bits of vote_kernel_queued:
512 = stop after the bitmap level, 1024 = queue the survivors but skip the peak_kmer gathers; results wrong with either): what a
better filter level could save at most.  Usage: ragged_vote_stages.py [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine
from localhgt_amd.synth import ragged_cuts
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
NC, CL = 13000, 1_000_000
with Engine(32, 3) as eng:
    eng.rng_seed(1); eng.coder_generate()
    eng.synth_reference_cuts(1, NC, CL, ragged_cuts(NC * CL))
    eng.synth_pairs(1, 2, NC, CL, 0, pairs, 150)
    eng.count_kmers()
    n = eng.ref_scan(0.1, 0.08, 300_000_000)
    print("raw peaks", n, eng.scan_info(), flush=True)
    for name, dbg in (("whole vote", 0), ("bitmap level only (bit 9)", 512), ("bitmap + queue, no gathers (bit 10)", 1024), ("whole vote", 0)):
        eng.set_debug(dbg)
        eng.work_stats(1)
        eng.vote()
        st = eng.work_stats(0)
        print(f"{name:40s} {eng.phase_ms(2):8.1f} ms  {eng.vote_info()}  survivors/pair {st['vote_hbm_probes'] / pairs:.1f}  revoted {st['vote_revoted_pairs']}", flush=True)
    eng.set_debug(0)
