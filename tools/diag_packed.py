#!/usr/bin/env python3
"""Where do the two resident forms of the reference (index / packed) disagree?  Exact scan in both forms, the single/trio bits
of every reference position compared chunk by chunk.  usage: diag_packed.py [n_contigs] [pairs] [sample_contigs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 13000
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
SC = int(sys.argv[3]) if len(sys.argv) > 3 else 200
CL, K, E = 1_000_000, 32, 3
CH = 1 << 29

with Engine(K, E) as e:
    e.rng_seed(1)
    e.coder_generate()
    e.synth_reference(1, NC, CL)
    e.synth_options(0, 20, SC)
    e.synth_pairs(1, 2, NC, CL, 0, PAIRS)
    e.synth_options(0, 20, 0)
    e.count_kmers()
    e.set_debug(8192)
    n_idx = e.ref_scan(0.1, 0.08, 300_000_000)
    n_pos = NC * CL
    ref = [e.flags_export(o, min(CH, n_pos - o)) & 0x7f for o in range(0, n_pos, CH)]
    print("index form:", n_idx, e.scan_info(), flush=True)
    e.set_reference_form(True)
    e.synth_reference(1, NC, CL)
    e.set_debug(8192)
    n_pk = e.ref_scan(0.1, 0.08, 300_000_000)
    print("packed form:", n_pk, e.scan_info(), flush=True)
    total = 0
    shown = 0
    per_contig = {}
    bits = [0] * 7
    for i, o in enumerate(range(0, n_pos, CH)):
        got = e.flags_export(o, min(CH, n_pos - o)) & 0x7f
        d = np.nonzero(got != ref[i])[0]
        total += d.size
        for b in range(7):
            bits[b] += int((((got[d] ^ ref[i][d]) >> b) & 1).sum())
        for x in d[:2000]:
            p = o + int(x)
            per_contig[p // CL] = per_contig.get(p // CL, 0) + 1
        for x in d[:8]:
            if shown < 64:
                p = o + int(x)
                print(f"  pos {p} = contig {p // CL} offset {p % CL} (word {p >> 5}, bit {p & 31}): index {ref[i][x]} packed {got[x]}")
                shown += 1
        if d.size:
            print(f"chunk {i}: {d.size} differing positions, offsets mod 1e6 min {int((o + d).min() % CL)} max {int(((o + d) % CL).max())}", flush=True)
    print("per-bit differences:", bits)
    print("differing positions:", total, "contigs touched (first 2000 per chunk):", len(per_contig), sorted(per_contig.items())[:20])
