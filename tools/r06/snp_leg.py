#!/usr/bin/env python3
"""the SNP 1 % deep focused leg (the reference's own read model: species20_snp0.01) with the shared-line-fill vote against the dense
generic kernel: phase times, line fills, digests.  usage: snp_leg.py [pairs] [sample_contigs] [snp_permille]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine
NC, CL, K, E = 13000, 1_000_000, 32, 3
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
sc = int(sys.argv[2]) if len(sys.argv) > 2 else 300
snp = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = Engine(K, E); eng.rng_seed(1); eng.coder_generate(); eng.set_reference_form(True)
eng.synth_reference(1, NC, CL)
eng.synth_options(snp, 20, sc); eng.synth_pairs(1, 2, NC, CL, 0, pairs); eng.synth_options(0, 20, 0)
eng.counts_clear(); eng.count_kmers()
print(f"{pairs} pairs from {sc} contigs, SNP {snp / 10} %: count {eng.phase_ms(0):.1f} ms", flush=True)
res = {}
for name, dbg in (("dense", 1 << 28), ("shared", 0), ("shared again", 0), ("shared, judge skipped", 1)):
    eng.set_debug(dbg)
    n = eng.ref_scan(0.1, 0.08, 300_000_000)
    eng.work_stats(1)
    t0 = time.time(); eng.vote(); eng.synchronize(); wall = time.time() - t0
    st = eng.work_stats(0)
    res[name] = (n, eng.digest(eng.DIGEST_VOTES))
    print(f"{name:24s} form {eng.vote_info()['form']:7s} vote {eng.phase_ms(2):8.1f} ms (wall {wall * 1e3:8.1f}), scan {eng.phase_ms(1):7.1f} ms, peaks {n}, "
          f"fetches {st['vote_shared_fetches']} = {st['vote_shared_fetches'] / pairs:.1f} per pair (+ {st['vote_shared_outside'] / pairs:.2f} outside the sets), pairs left to the generic kernel {st['vote_revoted_pairs']}, votes digest {res[name][1]}", flush=True)
eng.set_debug(0)
print("same votes:", res["dense"] == res["shared"] == res["shared again"])
