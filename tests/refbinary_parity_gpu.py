#!/usr/bin/env python3
"""The product against the REAL reference binary (oracle/_ref/extract_ref_z: the reference's own source compiled by
oracle/build_ref.sh with the zero-new[] shim, run with -t 1) at a size beyond the committed goldens: 20 x 1 Mbp, 400 000 pairs,
k = 32 (4 GiB count table, 16 GiB peak_kmer), on the GPU box's host.  Index bytes, genome.len.txt and the interval file must be
identical.  Kept out of the collected suite because the reference needs minutes.  usage: refbinary_parity_gpu.py [contigs] [pairs] [k] [threads]
threads > 1: the reference with its N threads run in creation order (oracle/_ref/libseqthreads.so preloaded: the schedule without races)
against the product's -t N emulation (`localhgt bkp` passes -t 10 by default)."""
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from localhgt_amd import extract_ref

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 20
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1
SHIM = os.path.join(ROOT, "oracle", "_ref", "libseqthreads.so")
E = 3
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "extract_ref_z")
if not os.path.exists(REF_BIN):
    sys.exit("oracle/_ref/extract_ref_z is not built (oracle/build_ref.sh needs /root/reference)")
tmp = tempfile.mkdtemp(prefix="lhgt_refbin_", dir="/tmp")
r, g = os.path.join(tmp, "ref"), os.path.join(tmp, "gpu")
os.makedirs(r)
fa, f1, f2 = bench.synth_files(r, K, E, NC, 1_000_000, PAIRS, 0)
shutil.copytree(r, g)
res = {}


def run_ref():
    t0 = time.time()
    env = dict(os.environ, LD_PRELOAD=SHIM) if T > 1 else dict(os.environ)
    res["p"] = subprocess.run([REF_BIN, "s.1.fq", "s.2.fq", "ref.fa", "i.txt", "0.1", "0.08", str(T), str(K), "3000000", str(E), "1", "1"],
                              cwd=r, capture_output=True, text=True, env=env)
    res["s"] = time.time() - t0


th = threading.Thread(target=run_ref)
t0 = time.time()
th.start()
a = extract_ref.Args(os.path.join(g, "s.1.fq"), os.path.join(g, "s.2.fq"), os.path.join(g, "ref.fa"), os.path.join(g, "i.txt"), 0.1, 0.08, T, K, 3_000_000, E, 1, 1.0)
rep = extract_ref.run(a, log=lambda *x: None, emulate_threads=T > 1)
print(f"product: {rep['total_s']:.2f} s, raw peaks {rep['n_peaks']}, filtered {rep['n_filtered']}", flush=True)
while th.is_alive():
    th.join(45)
    print(f"  reference binary running, {time.time() - t0:.0f} s", flush=True)
p = res["p"]
print(f"reference binary (-t {T}): rc {p.returncode}, {res['s']:.0f} s; " + " | ".join(l for l in p.stdout.splitlines() if "raw BKPs" in l or "Finish" in l)[:300], flush=True)
ok = p.returncode == 0
for name in ("i.txt", "ref.fa.genome.len.txt"):
    same = open(os.path.join(r, name), "rb").read() == open(os.path.join(g, name), "rb").read()
    print(f"{name}: {'IDENTICAL' if same else 'DIFFERENT'} ({os.path.getsize(os.path.join(r, name))} bytes)")
    ok = ok and same
if rep["ref_form"] == "packed":       # LHGT_REF_FORM=packed in the environment: the product neither reads nor writes an index file
    same = not os.path.exists(os.path.join(g, f"ref.fa.k{K}.h{E}.index.dat"))
    print(f"index file: none written by the product (packed reference, {rep['ref_resident_bytes']} bytes resident)")
else:
    x, y = (open(os.path.join(d, f"ref.fa.k{K}.h{E}.index.dat"), "rb").read() for d in (r, g))
    same = len(x) == len(y) and x[:1198] == y[:1198] and x[1200:] == y[1200:]     # bytes 1198-1199: the reference reads past its coder array
    print(f"index file: {'IDENTICAL' if same else 'DIFFERENT'} ({len(x)} bytes)")
shutil.rmtree(tmp, ignore_errors=True)
sys.exit(0 if ok and same else 1)
