#!/usr/bin/env python3
"""phase A on configs[2]'s reads, the key scatter's copy-out granule A/B in ONE process on ONE GPU (boxes differ by several per
cent): round 4's 16-byte groups (LHGT_PART_CG=8) against whole 128-byte lines (default, GeomBig::CG = 64), interleaved; the table
digests must agree with each other and with the compare-and-swap kernel's.  usage: phase_a_cg.py [pairs] [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with Engine(32, 3) as g:
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, 16, 1_000_000)
    g.synth_pairs(1, 2, 13000, 1_000_000, 0, pairs, 150)
    ms, dig = {}, {}
    for r in range(rounds + 1):
        variants = [("16-byte groups (r4)", {"LHGT_PART_CG": "8"}), ("whole lines (r6)", {})]
        for kv in os.environ.get("PHASE_A_VARIANTS", "").split(";"):      # more variants: "name=ENV:value,ENV:value;..."
            if kv:
                nm, envs = kv.split("=", 1)
                variants.append((nm, dict(e.split(":", 1) for e in envs.split(",") if e)))
        for name, env in variants:
            for key in ("LHGT_PART_CG", "LHGT_PART_X"): os.environ.pop(key, None)
            os.environ.update(env)
            g.counts_clear(); g.count_kmers()
            if r: ms.setdefault(name, []).append(g.phase_ms(0))
            dig[name] = g.digest(g.DIGEST_COUNTS)
    for name, v in ms.items():
        print(f"{name:20s} {min(v):7.1f} ms  ({' '.join(f'{x:.1f}' for x in v)})", flush=True)
    print("tables identical:", len(set(dig.values())) == 1, flush=True)
    if os.environ.get("PHASE_A_CAS", "1") != "0":
        g.set_count_mode(0)
        g.counts_clear(); g.count_kmers()
        print(f"compare-and-swap kernel: {g.phase_ms(0):.1f} ms; table identical: {g.digest(g.DIGEST_COUNTS) == dig['whole lines (r6)']}", flush=True)
