#!/usr/bin/env python3
"""phases B and C alone on configs[2] (packed reference): kernel times of 3 steps and the vote digest"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, 13000, 1_000_000)
    g.synth_pairs(1, 2, 13000, 1_000_000, 0, pairs, 150)
    g.counts_clear(); g.count_kmers()
    print(f"A {g.phase_ms(0):.1f} ms", flush=True)
    for i in range(3):
        n = g.ref_scan(0.1, 0.08, 300_000_000)
        g.vote()
        print(f"B {g.phase_ms(1):.1f} (probe kernel {g.phase_ms(3):.1f})  C {g.phase_ms(2):.1f} ms  peaks {n} votes {g.digest(g.DIGEST_VOTES)} peak_kmer {g.digest(g.DIGEST_PEAK_KMER)}", flush=True)
