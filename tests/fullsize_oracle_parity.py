#!/usr/bin/env python3
"""BASELINE configs[1] at FULL size against the CPU restatement (a parity check kept out of the collected suite only because it takes six minutes of host time): 1000 x 1 Mbp reference, 10 M pairs, k = 32, e = 3 as files; `extract_ref` on the GPU (index built in the first run, cached in
the second), oracle/lhgt_oracle.c on all host cores with that cached index; interval files byte for byte, raw peak and voted pair
counts.  usage: fullsize_oracle_parity.py [n_contigs] [n_pairs]   (prints a heartbeat while the CPU run is busy)

`--against-golden [out_dir]` (round 5): the same files through the product as `-t 1` and as `-t 10` (thread emulation: what
`localhgt bkp` passes), compared with what the REAL reference binary wrote for them -- tests/golden/configs1_full/, made by
make_golden.sh there in the build container from tests/synth_cpu.c's files, which must be the device generator's byte for byte
(inputs.sha256).  The product's interval files and the sha256 of the inputs are also left in out_dir (default
gpurun_out/configs1_full), so that the comparison can be repeated where the golden files are (tools/compare_configs1_golden.py)."""
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

GOLDEN = "--against-golden" in sys.argv
if GOLDEN:
    i = sys.argv.index("--against-golden")
    OUT = sys.argv[i + 1] if len(sys.argv) > i + 1 else os.path.join(ROOT, "gpurun_out", "configs1_full")
    del sys.argv[i:]
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
K, E = 32, 3


def against_golden():
    import hashlib
    import shutil
    gdir = os.path.join(ROOT, "tests", "golden", "configs1_full")
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="lhgt_full_", dir="/tmp")
    t0 = time.time()
    fa, f1, f2 = bench.synth_files(tmp, K, E, NC, 1_000_000, PAIRS, 0)
    print(f"inputs written in {time.time() - t0:.0f} s: {NC} x 1 Mbp, {PAIRS} pairs", flush=True)

    def sha(path):
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    lines = [f"{sha(p)}  {os.path.basename(p)}" for p in (fa, f1, f2)]
    open(os.path.join(OUT, "inputs.sha256"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines), flush=True)
    ok = True
    gold_in = os.path.join(gdir, "inputs.sha256")
    if os.path.exists(gold_in) and NC == 1000 and PAIRS == 10_000_000:
        same = open(gold_in).read().split() == "\n".join(lines).split()
        print("inputs vs the golden's inputs (tests/synth_cpu.c):", "IDENTICAL" if same else "DIFFERENT", flush=True)
        ok = ok and same
    for t in (1, 10):
        for rnd in range(2 if t == 1 else 1):          # the first -t 1 run builds the index, the rest reuse it
            a = extract_ref.Args(f1, f2, fa, os.path.join(tmp, f"gpu_t{t}.txt"), 0.1, 0.08, t, K, 300_000_000, E, 1, 1.0)
            rep = extract_ref.run(a, log=lambda *x: None)
        got = open(os.path.join(tmp, f"gpu_t{t}.txt"), "rb").read()
        open(os.path.join(OUT, f"interval_t{t}.txt"), "wb").write(got)
        print(f"product -t {t}: {rep['total_s']:.2f} s, emulated threads {rep['emulated_threads']}, raw peaks {rep['n_peaks']}, voted peaks {rep['n_filtered']}, "
              f"pairs {rep['pairs_kept']}, {got.count(10)} interval lines", flush=True)
        gold = os.path.join(gdir, f"interval_t{t}.txt")
        if os.path.exists(gold) and NC == 1000 and PAIRS == 10_000_000:
            same = open(gold, "rb").read() == got
            print(f"interval file -t {t} vs the reference binary's: {'IDENTICAL' if same else 'DIFFERENT'}", flush=True)
            ok = ok and same
    g = hashlib.sha256()
    with open(f"{fa}.k{K}.h{E}.index.dat", "rb") as f:
        head = bytearray(f.read(1200))
        head[1198:1200] = b"\0\0"       # the reference writes two bytes from behind its coder array there
        g.update(bytes(head))
        for blk in iter(lambda: f.read(1 << 24), b""):
            g.update(blk)
    lines = [f"{sha(fa + '.genome.len.txt')}  ref.fa.genome.len.txt", f"{g.hexdigest()}  ref.fa.k{K}.h{E}.index.dat (bytes 1198-1199 zeroed)"]
    gold_out = os.path.join(gdir, "outputs.sha256")
    if os.path.exists(gold_out) and NC == 1000 and PAIRS == 10_000_000:
        same = open(gold_out).read().split() == "\n".join(lines).split()
        print("genome.len.txt and index file vs the reference binary's:", "IDENTICAL" if same else "DIFFERENT", flush=True)
        ok = ok and same
    open(os.path.join(OUT, "outputs.sha256"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(0 if ok else 1)


if GOLDEN:
    against_golden()
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
