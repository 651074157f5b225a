import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from localhgt_amd.engine import Engine
NC, CL, K, E = 13000, 1_000_000, 32, 3
eng = Engine(K, E); eng.rng_seed(1); eng.coder_generate(); eng.set_reference_form(True)
eng.synth_reference(1, NC, CL)
eng.synth_pairs(1, 2, NC, CL, 0, int(2e9 / 300))
eng.counts_clear(); eng.count_kmers()
for i in range(3):
    eng.work_stats(1)
    n = eng.ref_scan(0.1, 0.08, 300_000_000)
    print(eng.scan_info(), eng.work_stats(), eng.phase_ms(1), flush=True)
