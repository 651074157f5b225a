#!/usr/bin/env python3
"""Generate the golden fixtures by running the REAL reference in the build container.

For every case in tests/cases.py: materialise the seeded inputs, run
oracle/_ref/extract_ref_z (the reference source compiled by oracle/build_ref.sh with the
zero-fill `new[]` shim, `-t 1`), then the reference's own scripts/get_bed_file.py, and store
    tests/golden/<case>/interval.txt, interval.txt.bed, genome.len.txt, meta.json
meta.json holds the sha256 of the three inputs and of the index file, the raw-peak count
printed by the reference and the `extracted ref length` line of get_bed_file.py.
Also captures the `bash pipeline.sh ...` command string of
scripts/infer_HGT_breakpoint.py:29 for a few flag sets (cmdline.json).

Runs only where /root/reference exists; its outputs are data (no reference source is stored).
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from cases import CASES, extract_ref_argv, materialise, sha256_file  # noqa: E402

REF = os.environ.get("LHGT_REFERENCE_DIR", "/root/reference")
BIN = os.path.join(ROOT, "oracle", "_ref", "extract_ref_z")


def run_case(case, keep_inputs=False):
    out = os.path.join(HERE, case.name)
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="lhgt_gold_") as tmp:
        fa, f1, f2 = materialise(case, tmp)
        interval = os.path.join(tmp, "interval.txt")
        argv = [BIN] + extract_ref_argv(case, f1, f2, fa, interval)
        runs = 2 if case.preexisting_index else 1
        env = dict(os.environ)
        if case.threads > 1:    # -t N contract: the reference's threads one after the other in creation order (oracle/seq_threads.c)
            env["LD_PRELOAD"] = os.path.join(ROOT, "oracle", "_ref", "libseqthreads.so")
        for _ in range(runs):
            res = subprocess.run(argv, capture_output=True, text=True, check=True, env=env)
        raw = int(re.findall(r"No\. of raw BKPs: (\d+)", res.stdout)[-1])
        index = f"{fa}.k{case.k}.h{case.e}.index.dat"
        meta = {
            "case": case.name, "notes": case.notes,
            "argv_tail": argv[5:],
            "sha256": {"ref.fa": sha256_file(fa), "s.1.fq": sha256_file(f1), "s.2.fq": sha256_file(f2),
                       "index.dat": sha256_file(index)},
            "index_bytes": os.path.getsize(index),
            "raw_peaks": raw,
            "index_preexisting": bool(case.preexisting_index),
            "threads": case.threads,
            "thread_log": re.findall(r">>> Thread: final read.*", res.stdout),
        }
        shutil.copy(interval, os.path.join(out, "interval.txt"))
        shutil.copy(fa + ".genome.len.txt", os.path.join(out, "genome.len.txt"))
        bed = subprocess.run([sys.executable, os.path.join(REF, "scripts", "get_bed_file.py"), fa, interval],
                             capture_output=True, text=True)
        meta["bed_returncode"] = bed.returncode
        meta["bed_stdout"] = bed.stdout
        if os.path.exists(interval + ".bed"):
            shutil.copy(interval + ".bed", os.path.join(out, "interval.txt.bed"))
        if keep_inputs:
            for p in (fa, f1, f2):
                with open(p, "rb") as src, gzip.GzipFile(os.path.join(out, os.path.basename(p) + ".gz"), "wb",
                                                         compresslevel=9, mtime=0) as dst:
                    shutil.copyfileobj(src, dst)
        with open(os.path.join(out, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(case.name, "raw_peaks", raw, "interval lines", sum(1 for _ in open(interval)), flush=True)


def capture_cmdlines():
    """Golden command strings from the reference's own argparse + Accept_Parameters (B:29)."""
    code = r'''
import sys, json, io, os, shutil, runpy, contextlib
scripts, sets = sys.argv[1], json.loads(sys.argv[2])
# run the reference driver as __main__ with the filesystem / PATH probes and os.system stubbed,
# so check_input -> refine_fastq -> is_tool -> get_order -> run all execute (B:123-184)
shutil.which = lambda name: "/usr/bin/" + name
os.path.isfile = lambda p: True
os.path.isdir = lambda p: True
out = {}
for name, args in sets.items():
    calls = []
    os.system = lambda cmd, calls=calls: calls.append(cmd) or 0
    sys.argv = ["/opt/x/infer_HGT_breakpoint.py"] + args
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            runpy.run_path(os.path.join(scripts, "infer_HGT_breakpoint.py"), run_name="__main__")
        except SystemExit:
            pass
    out[name] = {"args": args, "os_system": calls, "stdout": buf.getvalue()}
print(json.dumps(out))
'''
    sets = {
        "defaults": ["-r", "ref.fa", "--fq1", "a.1.fq", "--fq2", "a.2.fq"],
        "sample_half": ["-r", "ref.fa", "--fq1", "a.1.fq", "--fq2", "a.2.fq", "--sample", "0.5"],
        "all_flags": ["-r", "r.fa", "--fq1", "x_1.fq", "--fq2", "x_2.fq", "-s", "S1", "-o", "out", "-k", "24",
                      "-t", "4", "-e", "4", "-a", "0", "-q", "30", "--seed", "9", "--hit_ratio", "0.2",
                      "--match_ratio", "0.05", "--max_peak", "1000", "--sample", "3000000", "--read_info", "0"],
        "help": ["-h"],
    }
    res = subprocess.run([sys.executable, "-c", code, os.path.join(REF, "scripts"), json.dumps(sets)],
                         capture_output=True, text=True, check=True)
    with open(os.path.join(HERE, "cmdline.json"), "w") as f:
        json.dump(json.loads(res.stdout), f, indent=1, sort_keys=True)
    print("cmdline.json written")


def capture_count_diff():
    """count_diff_kmer.cpp on the committed k24_seed7 inputs, made reproducible by the two shims (time() fixed, threads in
    creation order): the two result lines it prints (C:45-48)"""
    tool = os.path.join(ROOT, "oracle", "_ref", "count_diff_kmer")
    pre = ":".join(os.path.join(ROOT, "oracle", "_ref", x) for x in ("libseqthreads.so", "libfixedtime.so"))
    out = []
    with tempfile.TemporaryDirectory(prefix="lhgt_gold_") as tmp:
        fa, f1, f2 = materialise(CASES["k24_seed7"], tmp)
        for k, ratio, t in ((12, 100, 1), (16, 100, 1), (20, 37, 5), (24, 80, 123456), (18, 3, 9)):
            res = subprocess.run([tool, f1, f2, str(k), str(ratio)], capture_output=True, text=True, check=True,
                                 env=dict(os.environ, LD_PRELOAD=pre, LHGT_FIXED_TIME=str(t)))
            lines = [ln for ln in res.stdout.splitlines() if ln.startswith("####") or re.match(r"^\d+\t\S+\t\S+$", ln) or ln.startswith("###kmer_is")]
            out.append({"case": "k24_seed7", "k": k, "ratio": ratio, "time": t, "lines": lines})
    with open(os.path.join(HERE, "count_diff_kmer.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("count_diff_kmer.json written")


if __name__ == "__main__":
    only = sys.argv[1:]
    capture_cmdlines()
    capture_count_diff()
    for name, case in CASES.items():
        if only and name not in only:
            continue
        run_case(case, keep_inputs=(name == "k24_seed7"))
