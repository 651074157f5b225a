#!/usr/bin/env python3
"""the deep focused sample (100 M pairs from 300 genomes of the 13 Gbase reference, packed): phase times and digests
usage: deep_focused_time.py [pairs] [genomes] [hit] [match] [ragged]   (ragged: the reference cut into ~118 k catalogue-like pieces)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
match = float(sys.argv[4]) if len(sys.argv) > 4 else 0.08
ragged = len(sys.argv) > 5 and sys.argv[5] == "ragged"
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    if ragged:
        from localhgt_amd.synth import ragged_cuts
        g.synth_reference_cuts(1, 13000, 1_000_000, ragged_cuts(13000 * 1_000_000))
    else:
        g.synth_reference(1, 13000, 1_000_000)
    g.synth_options(0, 20, nsamp)
    g.synth_pairs(1, 2, 13000, 1_000_000, 0, pairs, 150)
    g.counts_clear(); g.count_kmers()
    for i in range(2):
        n = g.ref_scan(hit, match, 300_000_000)
        g.vote()
        print(f"A {g.phase_ms(0):.1f} B {g.phase_ms(1):.1f} C {g.phase_ms(2):.1f} ms  peaks {n} votes {g.digest(g.DIGEST_VOTES)}", flush=True)
