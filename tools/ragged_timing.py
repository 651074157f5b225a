#!/usr/bin/env python3
"""Phase times of the headline workload on a RAGGED reference (the same 13 Gbase base stream cut into ~117 k contigs, 10 bases to
2 Mbp, three quarters of them shorter than 20 kb: tests/test_gpu_fullsize_uhgg.py::_ragged_cuts) next to the 13000 x 1 Mbp one.
Usage: ragged_timing.py [pairs]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from benchlib.legs import Workload
from test_gpu_fullsize_uhgg import _ragged_cuts, NC, CL
from localhgt_amd.engine import Engine
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
eng = Engine(32, 3)
eng.rng_seed(1); eng.coder_generate()
out = os.path.join(tempfile.gettempdir(), "ragged_iv.txt")
res = {}
for name in ("13000 x 1 Mbp", "ragged"):
    t0 = time.time()
    if name == "ragged":
        cuts = _ragged_cuts(NC * CL)
        eng.synth_reference_cuts(1, NC, CL, cuts)
        lens = np.diff(cuts.astype(np.int64))
        extra = {"contigs": int((lens > 32).sum()), "median_len": int(np.median(lens)), "shorter_than_a_tile": int((lens < 2000).sum())}
    else:
        eng.synth_reference(1, NC, CL)
        extra = {"contigs": NC}
    eng.synchronize()
    setup = time.time() - t0
    if eng.pairs_count() == 0:
        eng.synth_pairs(1, 2, NC, CL, 0, pairs)
    w = Workload(eng, None, 0, 1, False, out)
    dt, ms, n_peaks, nf = w.run(3, 1)
    res[name] = dict(extra, index_setup_s=round(setup, 2), ms_per_step=round(dt / 3 * 1e3, 1), count_A=round(ms[0], 1), scan_B=round(ms[1], 1),
                     vote_C=round(ms[2], 1), ref_flags=round(ms[3], 1), scan_form=eng.scan_info(), raw_peaks=n_peaks, filtered=nf)
    print(name, json.dumps(res[name]), flush=True)
eng.close()
