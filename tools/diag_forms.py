#!/usr/bin/env python3
"""Where do two forms of phase B disagree?  usage: diag_forms.py k n_contigs pairs sample_contigs debugA debugB [packed|index]
Scans the same table under two debug settings (include/localhgt_hip.h: lhgt_set_debug), twice each, and compares per contig the
counts of every flag bit; then the flag bytes of the first contigs that differ, position by position."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine

K, NC, PAIRS, SC, DA, DB = (int(x) for x in sys.argv[1:7])
PACKED = len(sys.argv) > 7 and sys.argv[7] == "packed"
CL, E = 1_000_000, 3
CH = 500 * CL


def per_contig(e):
    """(n_contigs, 7) counts of flag bits 0..6"""
    out = np.zeros((NC, 7), dtype=np.int64)
    for o in range(0, NC * CL, CH):
        n = min(CH, NC * CL - o)
        f = e.flags_export(o, n).reshape(n // CL, CL)
        for b in range(7):
            out[o // CL:o // CL + n // CL, b] = ((f >> b) & 1).sum(axis=1)
    return out


with Engine(K, E) as e:
    e.rng_seed(1)
    e.coder_generate()
    e.set_reference_form(PACKED)
    e.synth_reference(1, NC, CL)
    e.synth_options(0, 20, SC)
    e.synth_pairs(1, 2, NC, CL, 0, PAIRS)
    e.synth_options(0, 20, 0)
    e.count_kmers()
    runs = []
    for dbg in (DA, DB, DA, DB):
        e.set_debug(dbg)
        n = e.ref_scan(0.1, 0.08, 300_000_000)
        print("debug", dbg, "peaks", n, e.scan_info(), flush=True)
        runs.append(per_contig(e))
    for name, x, y in (("A vs A", runs[0], runs[2]), ("B vs B", runs[1], runs[3]), ("A vs B", runs[0], runs[1])):
        d = np.nonzero((x != y).any(axis=1))[0]
        print(name, ": contigs that differ", d.size, "first", d[:12].tolist(), "per-bit count differences", (x != y).sum(axis=0).tolist(), flush=True)
    d = np.nonzero((runs[0][:, 2:] != runs[1][:, 2:]).any(axis=1))[0]
    for c in d[:3]:
        e.set_debug(DA)
        e.ref_scan(0.1, 0.08, 300_000_000)
        fa = e.flags_export(int(c) * CL, CL) & 0x7f
        e.set_debug(DB)
        e.ref_scan(0.1, 0.08, 300_000_000)
        fb = e.flags_export(int(c) * CL, CL) & 0x7f
        w = np.nonzero((fa ^ fb) & 0x7c)[0]
        print(f"contig {c}: {w.size} positions differ in bits 2-6; tiles {sorted(set((w // 2000).tolist()))[:20]}")
        for x in w[:30]:
            print(f"   offset {x} (tile {x // 2000}, in tile {x % 2000}): A {fa[x]:#04x} B {fb[x]:#04x}")
