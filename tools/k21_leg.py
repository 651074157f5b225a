import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from benchlib.legs import Workload
from localhgt_amd.engine import Engine
nc, fp = 50_000, 25_000_000
for kk in (21,):
    with Engine(kk, 3) as e5:
        e5.rng_seed(1); e5.coder_generate(); e5.set_reference_form(True)
        e5.synth_reference(1, nc, 1_000_000)
        e5.synth_options(0, 20, 300)
        e5.synth_pairs(1, 2, nc, 1_000_000, 0, fp, 150)
        w = Workload(e5, None, 0, 1, False, '/tmp/iv.txt')
        dt, per_ms, n_peaks, nf = w.run(2, 1)
        print(kk, round(dt/2*1e3,1), [round(x,1) for x in per_ms], n_peaks, nf, e5.digest(e5.DIGEST_VOTES), flush=True)
        e5.set_debug(1)
        e5.vote(); print('vote without judge', round(e5.phase_ms(2),1), flush=True)
        e5.set_debug(0)
        e5.pairs_clear(); e5.synth_pairs(1, 2, nc, 1_000_000, 0, 2_000_000, 150)
        e5.counts_clear(); e5.count_kmers(); e5.ref_scan(0.1, 0.08, 300_000_000); e5.vote(); print('2M pairs vote', round(e5.phase_ms(2),1), flush=True)
