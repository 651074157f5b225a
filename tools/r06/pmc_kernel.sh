#!/bin/bash
# PMC counters of one kernel of a command: pmc_kernel.sh OUTNAME KERNEL_SUBSTR "CTR1 CTR2 ..." -- program args   (one --pmc pass, no tracing)
name=$1; kern=$2; ctrs=$3; shift; shift; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $out -o c -- "$@" > $out.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$out/**/*counter_collection.csv",recursive=True)
if not fs: print("no counter_collection.csv"); raise SystemExit
acc=collections.defaultdict(float); calls=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if "$kern" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); calls[r["Counter_Name"]]+=1
with open("$out.txt","a") as o:
    for k,v in sorted(acc.items()):
        line=f"$kern {k} total {v:.4g} over {calls[k]} dispatches = {v/calls[k]:.4g} per dispatch"
        print(line); o.write(line+"\n")
PY
