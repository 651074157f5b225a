#!/bin/bash
# per-kernel times of a command under rocprofv3 (kernel trace + stats, csv): prof_kernels.sh OUTNAME -- python3 script args...
name=$1; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- "$@" > $out.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
fs=glob.glob("$out/**/*kernel_stats.csv",recursive=True)
if not fs: print("no kernel_stats.csv under $out"); raise SystemExit
rows=list(csv.DictReader(open(fs[0])))
with open("$out.kernels.txt","w") as o:
    for r in rows[:25]:
        line=f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:10.2f} ms avg {float(r["AverageNs"])/1e6:9.3f} ms  {r.get("Percentage","")}'
        print(line); o.write(line+"\n")
PY
