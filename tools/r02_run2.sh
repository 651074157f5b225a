#!/bin/bash
# GPU run 2 of round 2: -t N goldens, probe-shape microbenchmark (+ fabric counters), ingest trace
out=gpurun_out/r02_b; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "golden" > $out/pytest_golden.log 2>&1; echo "pytest rc=$?" >> $out/rc.txt
tail -5 $out/pytest_golden.log
./tools/probe_shapes > $out/probe_shapes.txt 2>&1; echo "probe rc=$?" >> $out/rc.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$out/counters_list.txt 2>&1
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | tr " " "+")
  rm -rf /tmp/ps_$n
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/ps_$n -- $GRAFT_REPO_ROOT/tools/probe_shapes > /tmp/ps_$n.log 2>&1
  echo "pmc $n rc=$?" >> $GRAFT_REPO_ROOT/$out/rc.txt
  python3 - /tmp/ps_$n >> $GRAFT_REPO_ROOT/$out/probe_shapes_pmc.txt <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Counter_Name"])
        rows.setdefault(key, []).append(float(r["Counter_Value"]))
for (k, c), v in rows.items():
    print(f"{k[:60]:60s} {c:24s} n={len(v):3d} last={v[-1]:.4e}")
PY
done
cd $GRAFT_REPO_ROOT
LHGT_INGEST_TRACE=1 python tools/e2e_files.py > $out/e2e_trace.txt 2>&1
cat $out/rc.txt; head -60 $out/probe_shapes.txt; tail -20 $out/e2e_trace.txt
