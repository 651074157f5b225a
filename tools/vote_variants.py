#!/usr/bin/env python3
"""phase C of configs[2] (packed reference, 100 M pairs) under the A/B switches of vote_kernel_queued: ms per vote (best of 3) and the
vote / peak_kmer digests (must not move).  usage: vote_variants.py [pairs] [flag,flag,...]   (flags: lhgt_set_debug bits)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
flags = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1 << 17, 1 << 18, (1 << 17) | (1 << 18)]
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, 13000, 1_000_000)
    g.synth_pairs(1, 2, 13000, 1_000_000, 0, pairs, 150)
    g.counts_clear(); g.count_kmers()
    n = g.ref_scan(0.1, 0.08, 300_000_000)
    print(f"A {g.phase_ms(0):.1f} ms  B {g.phase_ms(1):.1f} ms  peaks {n}", flush=True)
    for f in flags + flags[:1]:
        g.set_debug(f)
        ms = []
        for _ in range(3):
            g.ref_scan(0.1, 0.08, 300_000_000)      # clears the votes
            g.vote()
            ms.append(g.phase_ms(2))
        print(f"debug {f:7d}: vote {min(ms):7.1f} ms (3 runs: {' '.join(f'{m:.1f}' for m in ms)})  votes {g.digest(g.DIGEST_VOTES)}", flush=True)
