#!/bin/bash
# the single-pass loader's host side under its knobs (io = pread | mmap, NOREUSE, threads): where a column's time goes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05c; mkdir -p $o
d=/tmp/ing; rm -rf $d; mkdir -p $d
python3 - <<'PY'
import sys, time
sys.path.insert(0, ".")
import bench
t0 = time.time()
bench.synth_files_sliced("/tmp/ing", 32, 3, 100, 1_000_000, int(__import__("os").environ.get("PAIRS", "32000000")), 0)
print(f"files written in {time.time() - t0:.0f} s", flush=True)
PY
run() { # name, env..., threads
  name=$1; shift; t=$1; shift
  for rep in 1 2; do
    echo "== $name threads=$t" >> $o/knobs.txt
    env "$@" LHGT_INGEST_TRACE=1 python3 tools/ingest_scaling.py --worker single $d/s.1.fq $d/s.2.fq 0 1 $t ${EMU:-1} /tmp >> $o/knobs.txt 2>&1
  done
}
: > $o/knobs.txt
: > $o/knobs.txt
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "v1 quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"; echo "affinity: $(taskset -p $$ 2>/dev/null)"; grep -i cpus_allowed_list /proc/self/status; cat /proc/self/cgroup | head -3; } > $o/cgroup.txt 2>&1
cat $o/cgroup.txt
for t in ${THREADS:-8 12 16 24 32 48}; do
  run pread $t LHGT_INGEST_IO=pread
  run mmap $t LHGT_INGEST_IO=mmap
done
grep -E "^==|one pass|line count" $o/knobs.txt | sed -e 's/columns of [0-9]* + [0-9]* bytes, //' -e 's/of \/tmp\/ing\/s.1.fq//' | cut -c1-260
echo "---- e2e (GPU loader), default thread count, then LHGT_INGEST_THREADS sweep" >> $o/knobs.txt
python3 - >> $o/knobs.txt 2>&1 <<'PY'
import os, sys, json, time
sys.path.insert(0, ".")
os.environ["LHGT_INGEST_TRACE"] = "1"
import bench
from localhgt_amd import extract_ref
from localhgt_amd.engine import Engine
d = "/tmp/ing"
fa = d + "/ref.fa"
def run(tag, threads):
    a = extract_ref.Args(d + "/s.1.fq", d + "/s.2.fq", fa, d + "/interval.txt", 0.1, 0.08, threads, 32, 300_000_000, 3, 1, 1.0)
    t0 = time.time()
    rep = extract_ref.run(a, log=lambda *x: None)
    rep["wall_s"] = time.time() - t0
    print(tag, json.dumps({k: round(v, 3) if isinstance(v, float) else v for k, v in rep.items() if k in ("pairs_kept", "reads_s", "index_s", "total_s", "count_kernel_ms", "count_s", "scan_s", "vote_s", "ingest_s", "wall_s")}),
          f"-> {rep['pairs_seen'] / rep['total_s'] / 1e6:.1f} M input pairs/s", flush=True)
run("index built in-run", 10)
for t in ("", "16", "24", "", "16", "24"):
    if t: os.environ["LHGT_INGEST_THREADS"] = t
    else: os.environ.pop("LHGT_INGEST_THREADS", None)
    for i in range(2):
        run(f"-t 10, ingest threads {t or 'default'}", 10)
PY
grep -E "^-t 10|^index built" $o/knobs.txt | cut -c1-300
