#!/usr/bin/env python3
"""phase A on configs[2]'s reads, variants interleaved in ONE process on ONE GPU (boxes differ by several per cent): round 3's sorted-tile
scatters (debug bit 16) against round 4's direct form; table digests must agree.  usage: phase_a_ab.py [pairs] [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, 16, 1_000_000)
    g.synth_pairs(1, 2, 13000, 1_000_000, 0, pairs, 150)
    ms, dig = {}, {}
    for r in range(rounds + 1):
        for name, dbg in (("sorted tiles (r3)", 65536), ("direct (r4)", 0)):
            g.set_debug(dbg)
            g.counts_clear(); g.count_kmers()
            if r: ms.setdefault(name, []).append(g.phase_ms(0))
            dig[name] = g.digest(g.DIGEST_COUNTS)
    for name, v in ms.items():
        print(f"{name:20s} {min(v):7.1f} ms  ({' '.join(f'{x:.1f}' for x in v)})", flush=True)
    print("tables identical:", len(set(dig.values())) == 1, flush=True)
