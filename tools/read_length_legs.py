#!/usr/bin/env python3
"""read length and the fast forms: the deep focused sample of bench.py (100 M pairs from 300 genomes of the 13 Gbase
reference) as 150-base reads, with 1 % of the pairs as 250-base reads, and as 250-base reads throughout: phase times of 2 steps.
Under `rocprofv3 --kernel-trace --stats` for the kernels.   usage: read_length_legs.py [pairs] [which: all | 150 | mixed | 250]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "all"
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, 13000, 1_000_000)
    g.synth_options(0, 20, 300)
    for name, mix, L in (("150", 0, 150), ("mixed", 10, 150), ("250", 0, 250)):
        if which not in ("all", name):
            continue
        g.pairs_clear()
        g.synth_read_mix(mix, 250 if mix else 0)
        g.synth_pairs(1, 2, 13000, 1_000_000, 0, pairs, L)
        g.synth_read_mix(0, 0)
        for i in range(2):
            g.counts_clear(); g.count_kmers()
            n = g.ref_scan(0.1, 0.08, 300_000_000)
            g.vote()
            ms = [g.phase_ms(i) for i in range(3)]
            print(f"{name:6s} A {ms[0]:.1f} B {ms[1]:.1f} C {ms[2]:.1f} ms = {pairs / sum(ms) / 1e3:.1f} M pairs/s (kernels)  peaks {n} votes {g.digest(g.DIGEST_VOTES)} {g.vote_info()['form']}", flush=True)
