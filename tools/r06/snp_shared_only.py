#!/usr/bin/env python3
"""the SNP leg's shared vote alone, N times (for counter passes): snp_shared_only.py [pairs] [sample_contigs] [votes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine
NC, CL, K, E = 13000, 1_000_000, 32, 3
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
sc = int(sys.argv[2]) if len(sys.argv) > 2 else 90
votes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng = Engine(K, E); eng.rng_seed(1); eng.coder_generate(); eng.set_reference_form(True)
eng.synth_reference(1, NC, CL)
eng.synth_options(10, 20, sc); eng.synth_pairs(1, 2, NC, CL, 0, pairs); eng.synth_options(0, 20, 0)
eng.counts_clear(); eng.count_kmers()
eng.set_debug((1 << 25) | int(os.environ.get("LHGT_DEBUG", "0")))
eng.ref_scan(0.1, 0.08, 300_000_000)
for _ in range(votes):
    eng.vote()
    print(f"vote {eng.phase_ms(2):.1f} ms, form {eng.vote_info()['form']}", flush=True)
