#!/usr/bin/env python3
"""stage ablation of vs_probe (LHGT_VS_ABLATE): the shared vote of the SNP leg with stages switched off, kernel time by phase_ms(2) minus the rest"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine
NC, CL, K, E = 13000, 1_000_000, 32, 3
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
sc = int(sys.argv[2]) if len(sys.argv) > 2 else 90
eng = Engine(K, E); eng.rng_seed(1); eng.coder_generate(); eng.set_reference_form(True)
eng.synth_reference(1, NC, CL)
eng.synth_options(10, 20, sc); eng.synth_pairs(1, 2, NC, CL, 0, pairs); eng.synth_options(0, 20, 0)
eng.counts_clear(); eng.count_kmers()
if len(sys.argv) > 3 and sys.argv[3] == "list":
    eng.slot_list(2)
else:
    eng.set_debug(1 << 25)
n = eng.ref_scan(0.1, 0.08, 300_000_000)
print("scan form", eng.scan_info()["form"], "list", eng.slot_list(), flush=True)
eng.vote()
for ab in (0, 16, 1 | 2 | 4, 2 | 4, 1 | 4, 4, 8, 2, 1, 0):
    os.environ["LHGT_VS_ABLATE"] = str(ab)
    eng.ref_scan(0.1, 0.08, 300_000_000)
    eng.vote(); eng.synchronize()
    print(f"ablate {ab:2d}: vote {eng.phase_ms(2):8.1f} ms", flush=True)
