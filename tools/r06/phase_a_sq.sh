#!/bin/bash
# SQ counters of phase A's three kernels on configs[2]'s reads (100 M pairs, one count): instruction mix, LDS bank
# conflicts, waits, occupancy.  Two --pmc passes (one run each; every kernel of the run is in the csv); output gpurun_out/r06/sq_phase_a.txt
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $out
rm -f $out/sq_phase_a.txt
export PHASE_A_ONLY=1
pass() {   # pass NAME "COUNTERS"
  d=$out/sqa_$1
  mkdir -p $d
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $2 --output-format csv -d $d -o c -- python3 $GRAFT_REPO_ROOT/tools/phase_a_time.py 100000000 > $d.log 2>&1 )
  python3 - "$d" "$2" >> $out/sq_phase_a.txt <<PY
import csv, glob, collections, sys
d, ctrs = sys.argv[1], sys.argv[2]
fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
print("# pass:", ctrs)
if not fs:
    print("no counter_collection.csv"); raise SystemExit
acc = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    for k in ("part_reads_direct", "part_keys16_direct", "part_apply2"):
        if k in r["Kernel_Name"]:
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:20s} {c:26s} total {v:.4e} over {calls[(k, c)]} dispatches")
PY
  rm -rf $d $d.log
}
pass 1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
pass 2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pass 3 "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_BUSY_CU_CYCLES"
cat $out/sq_phase_a.txt
