#!/bin/bash
out=gpurun_out/r02_d; mkdir -p $out
./tools/probe_shapes > $out/probe_shapes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d /tmp/pk -- $GRAFT_REPO_ROOT/tools/probe_shapes > /tmp/pk.log 2>&1
python3 - /tmp/pk > $GRAFT_REPO_ROOT/$out/probe_shapes_pmc.txt <<'PY'
import csv, glob, sys, collections, re
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = {}
for (d, k, g), v in sorted(rows.items()):
    last[(k, g)] = (d, v)     # keep the last of the three repetitions of every (kernel, grid)
for (k, g), (d, v) in sorted(last.items(), key=lambda kv: kv[1][0]):
    print(f"{d:4d} {k:28s} grid={g:>8s} " + " ".join(f"{c}={x:.4e}" for c, x in sorted(v.items())))
PY
cd $GRAFT_REPO_ROOT; cat $out/probe_shapes.txt; cat $out/probe_shapes_pmc.txt
