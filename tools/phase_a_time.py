#!/usr/bin/env python3
"""phase A alone on configs[2]'s reads: kernel time of the partitioned count (3 runs) and its whole table against the direct CAS kernel's
(device-side digest).  Usage: phase_a_time.py [pairs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, 16, 1_000_000)           # the reads only need the generator's base stream
    g.synth_pairs(1, 2, 13000, 1_000_000, 0, pairs, 150)
    g.set_debug(int(os.environ.get('PHASE_A_DEBUG', '0')))   # 65536 (bit 16): round 3's sorted-tile scatters
    for i in range(1 if os.environ.get('PHASE_A_ONLY') else 3):
        g.counts_clear(); g.count_kmers()
        print(f"partitioned count: {g.phase_ms(0):.1f} ms", flush=True)
    d1 = g.digest(g.DIGEST_COUNTS)
    if os.environ.get("PHASE_A_ONLY"):
        sys.exit(0)
    g.set_count_mode(0)
    g.counts_clear(); g.count_kmers()
    print(f"direct CAS count: {g.phase_ms(0):.1f} ms; tables identical: {g.digest(g.DIGEST_COUNTS) == d1}", flush=True)
