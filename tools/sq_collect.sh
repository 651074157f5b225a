#!/bin/bash
# SQ instruction-mix / stall counters for one bench.py run, one rocprofv3 pass per group; per-kernel sums printed on the box
# (the raw csv of the UHGG workload is too big to travel).  usage: tools/sq_collect.sh <outdir> <kernel substring> [bench args...]
out=$1; kern=$2; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_I8" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout 500 rocprofv3 --pmc $c --output-format csv -d /tmp/sq_$i -- python3 bench.py --no-cpu-baseline "$@" > "$out/pass$i.log" 2>&1
    echo "pass $i rc=$?"
    python3 - "$kern" /tmp/sq_$i >> "$out/summary.txt" <<'PY'
import csv, glob, sys, collections
kern, d = sys.argv[1], sys.argv[2]
fs = glob.glob(d + "/*/*_counter_collection.csv")
tot = collections.defaultdict(float); n = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in tot: print(f"{kern} {k} sum={tot[k]:.4e} dispatches={n[k]} per_dispatch={tot[k]/n[k]:.4e}")
PY
    rm -rf /tmp/sq_$i
done
cat "$out/summary.txt"
