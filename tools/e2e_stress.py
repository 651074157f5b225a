#!/usr/bin/env python3
"""the whole drop-in call (localhgt_amd.extract_ref.run) back to back on the same files -- 4 M pairs and a large file set, --sample 1 /
--sample 2e9 / packed reference -- in ONE process (how round 4 looked for the abort of the from-FASTQ legs, under rocgdb).
usage: e2e_stress.py big_pairs iterations"""
import sys, os, time, tempfile, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib.files import synth_files, synth_files_sliced
from localhgt_amd import extract_ref
n_big = int(sys.argv[1]); iters = int(sys.argv[2])
quiet = dict(device=0, log=lambda *x: None)
with tempfile.TemporaryDirectory(prefix="lhgt_e2e_") as tmp:
    fa, f1, f2 = synth_files(tmp, 32, 3, 100, 1_000_000, 4_000_000, 0)
    with tempfile.TemporaryDirectory(prefix="lhgt_e2e_") as tmp2:
        fb, g1, g2 = synth_files_sliced(tmp2, 32, 3, 100, 1_000_000, n_big, 0) if n_big else (fa, f1, f2)
        t0 = time.time()
        for it in range(iters):
            for (x1, x2, xa, t) in ((f1, f2, fa, tmp), (g1, g2, fb, tmp2)):
                for sample, kw in ((1.0, {}), (2e9, {}), (1.0, {"ref_form": "packed"})):
                    a = extract_ref.Args(x1, x2, xa, os.path.join(t, "interval.txt"), 0.1, 0.08, 10, 32, 300_000_000, 3, 1, sample)
                    r = extract_ref.run(a, **dict(quiet, **kw))
            print(it, round(time.time() - t0, 1), r["n_peaks"], flush=True)
