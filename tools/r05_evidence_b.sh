#!/bin/bash
# round 5, second GPU call: the GPU suite on the new loader (with per-test durations), from-files rates, host ingest scaling,
# kernel statistics of the CLI's default --sample regime
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05b
rm -rf $o; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_default -- python3 tools/default_sample_leg.py > $o/default_sample_leg.txt 2>&1
cp $(find $o/prof_default -name "*kernel_stats.csv" | head -1) $o/kernel_stats_default_sample.csv 2>/dev/null
rm -rf $o/prof_default
grep "^A " $o/default_sample_leg.txt
E2E_THREAD_SWEEP=24,48,96 timeout -k 10 400 python3 tools/e2e_big.py 32000000 100 1 > $o/e2e_stream.txt 2>&1
grep -v "staging + pinned" $o/e2e_stream.txt | cut -c1-360 | tail -16
timeout -k 10 500 python3 tools/ingest_scaling.py 32000000 $o/ingest_scaling.txt > $o/ingest_scaling.log 2>&1
tail -45 $o/ingest_scaling.log | cut -c1-200
if [ -n "$SUITE" ]; then
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=20 > $o/pytest_gpu.txt 2>&1
tail -30 $o/pytest_gpu.txt
fi
