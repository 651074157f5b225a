#!/bin/bash
# SQ instruction-mix / stall counters of the kernels whose names contain one of the given substrings, for any python command, one
# rocprofv3 pass per counter group (counters only: no trace domains).  usage: tools/sq_collect_cmd.sh <outfile> <sub1,sub2,...> <script.py> [args...]
out=$1; kerns=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > "$out"
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
         ${SQ_EXTRA:+"TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"} \
         ${SQ_EXTRA:+"TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"} \
         ${SQ_EXTRA:+"TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"}; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/sqc_$i -- python3 "$@" > /tmp/sqc_$i.log 2>&1
    echo "# pass $i rc=$? ($c)" >> "$out"
    python3 - "$kerns" /tmp/sqc_$i >> "$out" <<'PY'
import csv, glob, sys, collections
kerns, d = sys.argv[1].split(","), sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for k in kerns:
            if k in r["Kernel_Name"]:
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in kerns:
    for c in tot[k]:
        print(f"{k:28s} {c:28s} per_dispatch={tot[k][c] / n[k][c]:.4e} dispatches={n[k][c]}")
PY
    rm -rf /tmp/sqc_$i
done
cat "$out"
