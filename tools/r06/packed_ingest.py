#!/usr/bin/env python3
"""From files, text against packed (round 6): 32 M pairs as FASTQ (20 GB) and as a packed sample (4.7 GB), both in the page
cache.  One process: the whole load (read + upload + what the GPU does) of either.  N processes on the one GPU of the box, each loading
part i of N: the aggregate by the slowest rank, whole load and host side alone (the `read` seconds of the loader's trace) -- the host side
is what N ranks on N GPUs of one host would share.  usage: packed_ingest.py [pairs] [N ...]"""
import json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

if len(sys.argv) > 1 and sys.argv[1] == "--rank":
    # child: part `part` of `parts` of the packed sample (or of the FASTQ pair through the planned loader), -t 10 thread chunks
    kind, path1, path2, part, parts, ratio = sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7])
    os.environ["LHGT_INGEST_TRACE"] = "1"
    from localhgt_amd.engine import Engine
    from localhgt_amd import pack
    eng = Engine(32, 3)
    eng.rng_seed(1)
    eng.set_thread_emulation(10)
    if ratio < 100:
        eng.sampling_init(ratio)
    if kind != "packed":          # the line plans of both files, made outside the clock (a multi-rank run counts 1 / N of the lines per rank and exchanges them)
        ch1, ch2 = eng.fastq_pair_chunks(path1, path2)
        p1 = eng.fastq_plan_part(path1, 0, 1, chunk=ch1)
        p2 = eng.fastq_plan_part(path2, 0, 1, chunk=ch2)
    go = float(sys.argv[8])
    while time.time() < go:
        time.sleep(0.001)
    t0 = time.time()
    if kind == "packed":
        hdr = pack.read_header(path1)
        seen, kept = eng.pairs_load_packed(hdr, ratio, 10, part, parts)
    else:
        seen, kept = eng.pairs_load_fastq_planned(path1, path2, ratio, p1[:2], p2[:2], part, parts)
    eng.synchronize()
    print(json.dumps({"part": part, "s": time.time() - t0, "kept": kept}), flush=True)
    eng.close()
    sys.exit(0)

from benchlib.files import near_gpu, synth_files_sliced
from localhgt_amd import pack

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
Ns = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
os.system("nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null")
tmp = tempfile.mkdtemp(prefix="lhgt_pk_", dir="/tmp")
t0 = time.time()
with near_gpu(0):
    fa, f1, f2 = synth_files_sliced(tmp, 32, 3, 100, 1_000_000, pairs, 0)
print(f"FASTQ written in {time.time() - t0:.0f} s: 2 x {os.path.getsize(f1) / 1e9:.2f} GB", flush=True)
t0 = time.time()
out = os.path.join(tmp, "s.lhgp")
pack.pack(f1, f2, out, max_threads=10)
print(f"packed in {time.time() - t0:.1f} s: {os.path.getsize(out) / 1e9:.2f} GB", flush=True)


def run(kind, n, ratio):
    go = time.time() + 6.0 + 1.5 * n
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", kind, out if kind == "packed" else f1, f2, str(i), str(n), str(ratio), str(go)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(n)]
    recs, reads = [], []
    for p in procs:
        so, se = p.communicate()
        if p.returncode != 0:
            print(se[-2000:])
            raise SystemExit(f"{kind} rank failed")
        recs.append(json.loads(so.strip().splitlines()[-1]))
        m = re.search(r"read ([\d.]+)s", se) if kind == "packed" else None
        reads.append(float(m.group(1)) if m else 0.0)
    whole = max(r["s"] for r in recs)
    host = f"; host side alone (pread into pinned memory, max over ranks) {max(reads):.3f} s = {pairs / max(reads) / 1e6:7.1f} M pairs/s" if max(reads) > 0 else ""
    print(f"{kind:6s} {n} process(es), ratio {ratio:g} %: whole load {whole:.3f} s = {pairs / whole / 1e6:7.1f} M input pairs/s{host}; kept {sum(r['kept'] for r in recs)}", flush=True)


for ratio in (100.0, 6.67):
    run("fastq", 1, ratio)
    for n in Ns:
        run("packed", n, ratio)
    for n in Ns[1:]:
        run("fastq", n, ratio)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
