#!/usr/bin/env python3
"""One-off calibration of bench.py's cpu_baseline ("port"): the REAL reference binary (oracle/_ref/extract_ref_raw, built from
/root/reference by oracle/build_ref.sh; -O2) and the CPU restatement on the same generated files, on this box's host cores.
The reference prints whole seconds per phase, so the sample is sized for tens of seconds.  Usage: ref_cpu_timing.py [pairs] [threads]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
ref_bin = os.path.join(ROOT, "oracle", "_ref", "extract_ref_raw")
with tempfile.TemporaryDirectory(prefix="lhgt_refcpu_") as tmp:
    fa, f1, f2 = bench.synth_files(tmp, 32, 3, 20, 1_000_000, n_pairs, 0)
    # index first (single-threaded read_ref, not part of the rate): run once with the reads, keep the index
    t0 = time.time()
    res = subprocess.run([ref_bin, f1, f2, fa, os.path.join(tmp, "i0.txt"), "0.1", "0.08", str(threads), "32", "3000000", "3", "1", "1"],
                         capture_output=True, text=True)
    t_first = time.time() - t0
    t0 = time.time()
    res = subprocess.run([ref_bin, f1, f2, fa, os.path.join(tmp, "i1.txt"), "0.1", "0.08", str(threads), "32", "3000000", "3", "1", "1"],
                         capture_output=True, text=True)
    t_cached = time.time() - t0
    out = res.stdout
    count_s = re.findall(r"K-mer counting is finished. It costs (\d+) seconds", out)
    total_s = re.findall(r"Finish with time:\s*(\d+)", out)
    print(f"reference binary, -t {threads}, {n_pairs} pairs vs 20 x 1 Mbp, k=32 e=3: wall {t_cached:.1f} s with a cached index "
          f"({t_first:.1f} s building it); its own clock: counting (incl. 4 GiB memset + 50 M rand()) {count_s} s, whole run {total_s} s")
    print(f"=> {n_pairs / t_cached / 1e6:.4f} M pairs/s whole process, cached index")
    print("\n".join(l for l in out.splitlines() if "Slided" in l or "cost" in l.lower())[:1500])
    print("(bench.py itself now times the reference binary and the port on every run: tools/benchlib/cpu.py)")
