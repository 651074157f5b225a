#!/usr/bin/env python3
"""the 13 Gbase reference under the pipeline's default --sample 2000000000 (6.67 M of 100 M synthetic pairs kept; the sample is half
of the catalogue): phase times of 2 steps.  Under `rocprofv3 --kernel-trace --stats` for the kernels of phase B.
usage: default_sample_leg.py [sample_contigs (0 = half the reference)]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
nsamp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kept = int(2e9 / 300)
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, 13000, 1_000_000)
    g.synth_options(0, 20, nsamp)
    g.synth_pairs(1, 2, 13000, 1_000_000, 0, kept, 150)
    g.counts_clear(); g.count_kmers()
    for i in range(2):
        n = g.ref_scan(0.1, 0.08, 300_000_000)
        g.vote()
        print(f"A {g.phase_ms(0):.1f} B {g.phase_ms(1):.1f} C {g.phase_ms(2):.1f} ms  peaks {n} votes {g.digest(g.DIGEST_VOTES)} {g.scan_info()}", flush=True)
