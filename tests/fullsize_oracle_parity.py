#!/usr/bin/env python3
"""BASELINE configs[1] at FULL size against the CPU restatement (a parity check kept out of the collected suite only because it takes six minutes of host time): 1000 x 1 Mbp reference, 10 M pairs, k = 32, e = 3 as files; `extract_ref` on the GPU (index built in the first run, cached in
the second), oracle/lhgt_oracle.c on all host cores with that cached index; interval files byte for byte, raw peak and voted pair
counts.  usage: fullsize_oracle_parity.py [n_contigs] [n_pairs]   (prints a heartbeat while the CPU run is busy)"""
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # this script lives there: the oracle is test infrastructure
import bench
import oracle_api
from conftest import build_oracle
from localhgt_amd import extract_ref

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
K, E = 32, 3
tmp = tempfile.mkdtemp(prefix="lhgt_full_", dir="/tmp")
t0 = time.time()
fa, f1, f2 = bench.synth_files(tmp, K, E, NC, 1_000_000, PAIRS, 0)
print(f"inputs written in {time.time() - t0:.0f} s: {NC} x 1 Mbp, {PAIRS} pairs", flush=True)
reps = []
for run in range(2):
    a = extract_ref.Args(f1, f2, fa, os.path.join(tmp, f"gpu{run}.txt"), 0.1, 0.08, 10, K, 300_000_000, E, 1, 1.0)
    reps.append(extract_ref.run(a, log=lambda *x: None, emulate_threads=False))      # the CPU restatement below is the -t 1 one
    print("GPU run", run, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in reps[-1].items()}), flush=True)
orc = oracle_api.Oracle(build_oracle())
orc.set_pretouch(True)
res = {}


def cpu():
    res["rc"], res["rep"] = orc.run(f1, f2, fa, os.path.join(tmp, "cpu.txt"), 0.1, 0.08, os.cpu_count() or 1, K, 300_000_000, E, 1, 1.0)


th = threading.Thread(target=cpu)
t0 = time.time()
th.start()
while th.is_alive():
    th.join(45)
    print(f"  CPU restatement running, {time.time() - t0:.0f} s", flush=True)
rep = res["rep"]
print(f"CPU restatement ({os.cpu_count()} threads, cached index): rc {res['rc']}, A {rep.t_count:.0f} s, B {rep.t_scan:.0f} s, C {rep.t_vote:.0f} s, "
      f"raw peaks {rep.n_peaks}, pairs voted {rep.pairs_voted}", flush=True)
gpu = open(os.path.join(tmp, "gpu1.txt"), "rb").read()
cpu_b = open(os.path.join(tmp, "cpu.txt"), "rb").read()
same = gpu == cpu_b and reps[1]["n_peaks"] == rep.n_peaks and reps[1]["pairs_kept"] == rep.pairs_voted
print(f"interval files: GPU {len(gpu)} bytes / {gpu.count(10)} lines, CPU {len(cpu_b)} bytes -- {'IDENTICAL' if gpu == cpu_b else 'DIFFERENT'}; "
      f"raw peaks {reps[1]['n_peaks']} vs {rep.n_peaks}; pairs {reps[1]['pairs_kept']} vs {rep.pairs_voted}")
print("first run (index built in-run) equals second (cached):", open(os.path.join(tmp, "gpu0.txt"), "rb").read() == gpu)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
sys.exit(0 if same else 1)
