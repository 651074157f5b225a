#!/usr/bin/env python3
"""`extract_ref` from FASTQ files at a size where the reads, not the fixed costs, decide: n_pairs pairs written in
slices (device generator -> text), then the whole call with a cached index, with the packed reference, with -t 1 and with the
default -t 10 (thread emulation), LHGT_INGEST_TRACE on.  Usage: e2e_big.py [n_pairs] [n_contigs] [sample]"""
import json, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LHGT_INGEST_TRACE"] = "1"
from localhgt_amd.engine import Engine
from localhgt_amd import extract_ref
import bench

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
n_contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sample = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
CL, K, E, SL = 1_000_000, 32, 3, 4_000_000
os.system("free -g | head -2; df -h /tmp | tail -1; nproc")
tmp = tempfile.mkdtemp(prefix="lhgt_e2e_", dir="/tmp")
fa, f1, f2 = (os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq"))
t0 = time.time()
sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib.files import near_gpu
# E2E_FILES=far|anywhere: write the files from the OTHER socket / wherever the scheduler puts this process (default: the GPU's socket)
_where = os.environ.get("E2E_FILES", "near")
_ctx = near_gpu(0) if _where == "near" else near_gpu(0, away=True) if _where == "far" else near_gpu(-1)
_ctx.__enter__()
with Engine(K, E) as eng:
    eng.rng_seed(1); eng.coder_generate()
    ref = eng.synth_reference(1, n_contigs, CL, want_host=True)
    bench.write_fasta(fa, ref, n_contigs, CL)
    del ref
    for path in (f1, f2):
        open(path, "wb").close()
    for p0 in range(0, n_pairs, SL):
        n = min(SL, n_pairs - p0)
        eng.pairs_clear()
        m1, m2 = eng.synth_pairs(1, 2, n_contigs, CL, p0, n, 150, want_host=True)
        for path, m, suf in ((f1, m1, "1"), (f2, m2, "2")):
            part = path + ".part"
            bench.write_fastq(part, m, n, 150, suf)          # ids restart per slice: the path never looks at them beyond the first
            with open(path, "ab") as dst, open(part, "rb") as src:
                shutil.copyfileobj(src, dst, 1 << 24)
            os.remove(part)
_ctx.__exit__(None, None, None)
print(f"inputs written in {time.time() - t0:.0f} s ({_where} the GPU's NUMA node): fasta {os.path.getsize(fa) / 1e6:.0f} MB, fastq 2 x {os.path.getsize(f1) / 1e9:.2f} GB", flush=True)


def run(tag, threads, **kw):
    a = extract_ref.Args(f1, f2, fa, os.path.join(tmp, "interval.txt"), 0.1, 0.08, threads, K, 300_000_000, E, 1, sample)
    t0 = time.time()
    rep = extract_ref.run(a, log=lambda *x: None, **kw)
    rep["wall_s"] = time.time() - t0
    keep = ("pairs_seen", "pairs_kept", "n_peaks", "n_filtered", "ratio", "index_s", "reads_s", "count_s", "scan_s", "vote_s", "total_s", "wall_s", "emulated_threads")
    print(tag, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in rep.items() if k in keep}),
          f"-> {rep['pairs_seen'] / rep['total_s'] / 1e6:.1f} M input pairs/s", flush=True)
    return rep


if os.environ.get("E2E_KNOBS"):      # "name=VAR:val,VAR:val;name2=..." environment variants, each run twice with a cached index
    os.system("uname -r")
    run("index built in-run, -t 1", 1)
    for spec in os.environ["E2E_KNOBS"].split(";"):
        name, _, kv = spec.partition("=")
        sets = dict(x.split(":") for x in kv.split(",") if x)
        for kk in ("LHGT_INGEST_THREADS", "LHGT_MMAP_ADVICE", "LHGT_MMAP_POPULATE", "LHGT_INGEST_CHUNK_BYTES", "LHGT_ONE_COPY_STREAM", "LHGT_INGEST_STREAM", "LHGT_INGEST_IO", "LHGT_INGEST_NUMA"):
            os.environ.pop(kk, None)
        os.environ.update(sets)
        for i in range(2):
            run(f"cached, {name}", int(os.environ.get("E2E_T", "10")))
    shutil.rmtree(tmp)
    sys.exit(0)
if os.environ.get("E2E_THREAD_SWEEP"):
    run("index built in-run, -t 1", 1)
    for t in (int(x) for x in os.environ["E2E_THREAD_SWEEP"].split(",")):
        os.environ["LHGT_INGEST_THREADS"] = str(t)
        for i in range(2):
            run(f"index cached, -t 1, {t} ingest threads", 1)
    shutil.rmtree(tmp)
    sys.exit(0)
if os.environ.get("E2E_ONLY_PACKED"):       # catalogue-scale reference: no 12-bytes-per-base index file is written or read
    for i in range(2):
        run("packed reference, -t 10 (the CLI default)", 10, ref_form="packed")
    run("packed reference, -t 1", 1, ref_form="packed")
    shutil.rmtree(tmp)
    sys.exit(0)
run("index built in-run, -t 1", 1)
for i in range(2):
    run("index cached, -t 1", 1)
run("index cached, -t 10 (thread emulation, the CLI default)", 10)
run("packed reference, -t 1", 1, ref_form="packed")
run("packed reference, -t 10", 10, ref_form="packed")
shutil.rmtree(tmp)
