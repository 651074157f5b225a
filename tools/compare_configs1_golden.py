#!/usr/bin/env python3
"""BASELINE configs[1] at full size: the product's outputs on the GPU box (gpurun_out/configs1_full/, left there by
tests/fullsize_oracle_parity.py --against-golden) against the REAL reference binary's on the same bytes (tests/golden/configs1_full/,
made by make_golden.sh there).  Writes the comparison to stdout.  usage: compare_configs1_golden.py [product_dir] [golden_dir]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prod = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "configs1_full")
gold = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "configs1_full")
ok = True


def same(name, what):
    global ok
    a, b = os.path.join(prod, name), os.path.join(gold, name)
    if not (os.path.exists(a) and os.path.exists(b)):
        print(f"{what}: MISSING ({'product' if not os.path.exists(a) else 'golden'} side)")
        ok = False
        return
    x, y = open(a, "rb").read(), open(b, "rb").read()
    if name.endswith(".sha256"):                     # the digests in order; the labels behind them are for the reader
        x, y = (b" ".join(line.split()[0] for line in z.splitlines() if line.strip()) for z in (x, y))
    print(f"{what}: {'IDENTICAL' if x == y else 'DIFFERENT'}" + (f" ({x.count(10)} lines, {len(x)} bytes)" if name.endswith(".txt") else ""))
    ok = ok and x == y


print("BASELINE configs[1] at full size -- 1000 x 1 Mbp, 10 M 150 bp pairs, k = 32, e = 3, seed 1, --sample 1, max_peak 300000000")
print("reference: oracle/_ref/extract_ref_z (src/extract_ref_normal_peak.cpp + the zero-new[] shim), run in the build container by")
print("tests/golden/configs1_full/make_golden.sh; product: bin/extract_ref's entry point on an MI355X box (tests/fullsize_oracle_parity.py")
print("--against-golden).  The inputs are made by tests/synth_cpu.c here and by the device generator there:")
same("inputs.sha256", "sha256 of ref.fa, s.1.fq, s.2.fq")
same("interval_t1.txt", "interval file, -t 1")
same("interval_t10.txt", "interval file, -t 10 (reference: its threads in creation order; product: the thread emulation)")
same("outputs.sha256", "sha256 of genome.len.txt and of the 12 GB index file (bytes 1198-1199 zeroed)")
meta = os.path.join(gold, "meta.txt")
if os.path.exists(meta):
    print(open(meta).read().rstrip())
sys.exit(0 if ok else 1)
