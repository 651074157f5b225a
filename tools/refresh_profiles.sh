#!/bin/bash
# Re-create the evidence under profiles/<name>/ on the GPU box: kernel-trace stats + the bench line of the same run, PMC
# summaries (one pass per counter group, summarised here because the raw CSVs are too big to travel), plain bench runs.
# usage (from the repo root on the GPU box): tools/refresh_profiles.sh gpurun_out/<name>
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in uhgg 1g; do
    rm -rf /tmp/kt_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /tmp/kt_$w.log 2>/dev/null
    cp $(ls /tmp/kt_$w/*/*kernel_stats.csv | head -1) "$out/kernel_stats_$w.csv"
    grep '^{' /tmp/kt_$w.log | tail -1 > "$out/bench_${w}_under_rocprof.json"
    rm -rf /tmp/pmc_$w
    tools/pmc_collect.sh /tmp/pmc_$w --workload $w --steps 1 --warmup 0 > "$out/pmc_$w.log" 2>&1
    python3 tools/pmc_summarise.py /tmp/pmc_$w > "$out/pmc_table_$w.txt" 2>&1
    cp /tmp/pmc_$w/summary.json "$out/pmc_summary_$w.json"
    rm -rf /tmp/pmc_$w /tmp/kt_$w
    python3 bench.py --workload $w 2>/dev/null | grep '^{' | tail -1 > "$out/bench_${w}_run.json"
done
ls -la "$out"
