#!/bin/bash
out=gpurun_out/r02_c; mkdir -p $out
./tools/probe_shapes kinds > $out/probe_kinds.txt 2>&1; echo "rc=$?" >> $out/probe_kinds.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum --output-format csv -d /tmp/pk -- $GRAFT_REPO_ROOT/tools/probe_shapes kinds > /tmp/pk.log 2>&1
python3 - /tmp/pk > $GRAFT_REPO_ROOT/$out/probe_kinds_pmc.txt <<'PY'
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:40]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), v in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
    print(d, k, {c: f"{x:.3e}" for c, x in v.items()})
PY
cd $GRAFT_REPO_ROOT; cat $out/probe_kinds.txt; tail -50 $out/probe_kinds_pmc.txt
