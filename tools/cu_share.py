#!/usr/bin/env python3
"""Per-phase kernel time against the share of the GPU's CUs a context may use, and what two contexts with complementary shares
make of it (overlap phases across samples).  configs[2]: 13000 x 1 Mbp resident as packed bases, 100 M pairs.
CU masks: bit i of hipExtStreamCreateWithCUMask is CU i / 8 of XCD i % 8 (tools/cu_mask_probe.hip), so the first n bits are
n / 8 CUs of every XCD.  Usage: cu_share.py [pairs] [contigs]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 13000
CL = 1_000_000
K, E = 32, 3


def make(seed):
    g = Engine(K, E)
    g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
    g.synth_reference(1, NC, CL)
    g.synth_pairs(1, seed, NC, CL, 0, pairs, 150)
    return g


def step(g):
    g.counts_clear(); g.count_kmers()
    a = g.phase_ms(0)
    n = g.ref_scan(0.1, 0.08, 300_000_000)
    b, bk = g.phase_ms(1), g.phase_ms(3)
    g.vote()
    return dict(A=round(a, 1), B=round(b, 1), B_probe_kernel=round(bk, 1), C=round(g.phase_ms(2), 1), peaks=n, votes=g.digest(g.DIGEST_VOTES)[0])


engs = [make(2), make(3)]
for g in engs:
    step(g)                      # warm-up
print("== one context alone, CUs it may use (n / 8 of every XCD)", flush=True)
alone = {}
for n in (256, 224, 192, 160, 128, 96, 64, 32):
    engs[0].set_cu_mask(range(n) if n < 256 else None)
    r = step(engs[0])
    alone[n] = r
    print(f"{n:4d} CUs: A {r['A']:7.1f}  B {r['B']:7.1f} (probe kernel {r['B_probe_kernel']:7.1f})  C {r['C']:7.1f}   sum {r['A'] + r['B'] + r['C']:7.1f} ms", flush=True)
engs[0].set_cu_mask(None)
ref = [step(g) for g in engs]

print("== two contexts, complementary CU sets: context 0 loops phase A on n CUs, context 1 loops phases B+C on 256 - n", flush=True)


def both(n_a, n_samples=4, masked=True):
    """pipeline: a sample's A on the A-set (any context), its B+C on the rest; one lock per kind of phase"""
    la, lb = threading.Lock(), threading.Lock()
    recs = [[], []]

    def worker(i):
        g = engs[i]
        for _ in range(n_samples // 2):
            with la:
                if masked:
                    g.set_cu_mask(range(n_a))
                g.counts_clear(); g.count_kmers()
                a = g.phase_ms(0)
            with lb:
                if masked:
                    g.set_cu_mask(range(n_a, 256))
                n = g.ref_scan(0.1, 0.08, 300_000_000)
                b = g.phase_ms(1)
                g.vote()
                recs[i].append((a, b, g.phase_ms(2), n, g.digest(g.DIGEST_VOTES)[0]))
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.time() - t0
    for g in engs:
        g.set_cu_mask(None)
    ok = all(r[3] == ref[i]["peaks"] and r[4] == ref[i]["votes"] for i in range(2) for r in recs[i])
    m = lambda j: round(sum(r[j] for rr in recs for r in rr) / sum(len(rr) for rr in recs), 1)
    return dict(n_a=n_a, s=round(dt, 3), ms_per_sample=round(dt / n_samples * 1e3, 1), A=m(0), B=m(1), C=m(2), same_results=ok)


serial_ms = ref[0]["A"] + ref[0]["B"] + ref[0]["C"]
print(f"serial: {serial_ms:.1f} ms per sample (A {ref[0]['A']} B {ref[0]['B']} C {ref[0]['C']})", flush=True)
r = both(256, masked=False)
print("no masks (both contexts on all CUs):", json.dumps(r), f"gain {serial_ms / r['ms_per_sample']:.3f}", flush=True)
best = None
for n_a in (64, 96, 128, 160, 192):
    r = both(n_a)
    print(json.dumps(r), f"gain {serial_ms / r['ms_per_sample']:.3f}", flush=True)
    if best is None or r["ms_per_sample"] < best["ms_per_sample"]:
        best = r
print("best:", json.dumps(best), flush=True)
for g in engs:
    g.close()
