#!/usr/bin/env python3
"""the CLI's default --sample regime on the half-of-the-catalogue sample (6.67 M of 100 M pairs kept): phase B by kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine
NC, CL, K, E = 13000, 1_000_000, 32, 3
eng = Engine(K, E); eng.rng_seed(1); eng.coder_generate(); eng.set_reference_form(True)
eng.synth_reference(1, NC, CL)
kept = int(2e9 / 300)
eng.synth_pairs(1, 2, NC, CL, 0, kept)
eng.counts_clear(); eng.count_kmers()
for i in range(int(os.environ.get("LEG_SCANS", "3"))):
    n = eng.ref_scan(0.1, 0.08, 300_000_000)
    eng.vote()
    print(f"scan {eng.phase_ms(1):.1f} ms ({eng.scan_info()['form']}), vote {eng.phase_ms(2):.1f} ms ({eng.vote_info()['form']}), peaks {n}, registry {eng.registry_info()}, peak_kmer digest {eng.digest(eng.DIGEST_PEAK_KMER)}, votes {eng.digest(eng.DIGEST_VOTES)}", flush=True)
