#!/bin/bash
# Collect HBM-traffic PMC counters for bench.py, one rocprofv3 pass per counter group
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -> separate passes).
# usage: tools/pmc_collect.sh <outdir> [bench args...]
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr " " "+")
    timeout 500 rocprofv3 --pmc $c --output-format csv -d "$out/$n" -- python3 bench.py --no-cpu-baseline "$@" > "$out/$n.log" 2>&1
    echo "$n rc=$?"
done
