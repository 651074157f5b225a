#!/bin/bash
# Re-create the evidence under profiles/<name>/ on the GPU box: kernel-trace stats + the bench line of the same run (its
# roofline.launch_ms must agree with the CSV's average), the PMC summaries bench.py itself collects (--pmc-out), plain bench runs.
# usage (from the repo root on the GPU box): tools/refresh_profiles.sh gpurun_out/<name>
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in uhgg 1g; do
    rm -rf /tmp/kt_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc --no-verify > /tmp/kt_$w.log 2>/dev/null
    cp $(ls /tmp/kt_$w/*/*kernel_stats.csv | head -1) "$out/kernel_stats_$w.csv"
    grep '^{' /tmp/kt_$w.log | tail -1 > "$out/bench_${w}_under_rocprof.json"
    rm -rf /tmp/kt_$w
    python3 bench.py --workload $w --no-extras --no-cpu-baseline --pmc-out "$out/pmc_live_$w.json" 2>/dev/null | grep '^{' | tail -1 > "$out/bench_${w}_run.json"
done
ls -la "$out"
