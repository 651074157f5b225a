#!/usr/bin/env python3
"""phase A of the direct form (k = 32, e = 3, reads of <= 159 bases) against the compare-and-swap kernel over random shapes: read counts
1 .. 70 000, lengths 0 .. 159, N rates, hot k-mers that overflow tile rows and pieces, mates not counted.  usage: phase_a_soak.py first_seed end_seed"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from localhgt_amd.engine import Engine
acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
def pairs_arrays(r1, r2):
    s1 = np.frombuffer(b"".join(r1), dtype=np.uint8); s2 = np.frombuffer(b"".join(r2), dtype=np.uint8)
    o1 = np.cumsum([0] + [len(r) for r in r1]).astype(np.uint64); o2 = np.cumsum([0] + [len(r) for r in r2]).astype(np.uint64)
    return s1, o1, s2, o2
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    pn = float(rng.choice([0.0, 0.002, 0.02]))
    p = [(1 - pn) / 4] * 4 + [pn]
    lo = int(rng.choice([0, 20, 32, 100, 150])); hi = int(rng.choice([lo + 1, 151, 160]))
    hi = max(hi, lo + 1)
    def rd(n):
        return [acgt[rng.choice(5, size=int(rng.integers(lo, hi)), p=p)].tobytes() for _ in range(n)]
    n = int(rng.choice([1, 63, 1000, 30001, 70000]))
    hot = acgt[rng.choice(4, size=150)].tobytes()
    nh = int(rng.choice([0, 0, 300, 2000]))
    r1 = rd(n) + [b"A" * 150, b"ACAC" * 37, hot][:3] * nh
    r2 = rd(n) + [b"T" * 150, b"GTGT" * 37, hot][:3] * nh
    c2 = (rng.random(len(r1)) < float(rng.choice([1.0, 0.9, 0.5]))).astype(np.uint8)
    got = []
    for mode, dbg in ((0, 0), (1, 0)):
        with Engine(32, 3) as eng:
            eng.rng_seed(seed + 5); eng.coder_generate()
            eng.set_count_mode(mode); eng.set_debug(dbg)
            eng.pairs_append(*pairs_arrays(r1, r2), count_mate2=c2)
            eng.count_kmers()
            got.append((eng.digest(eng.DIGEST_COUNTS), tuple(int(x) for x in eng.counts_histogram())))
    ok = got[0] == got[1]
    print(seed, "n", n, "len", lo, hi, "pN", pn, "hot", nh, "OK" if ok else "MISMATCH", got[0][1][1:], flush=True)
    assert ok
