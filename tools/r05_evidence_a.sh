#!/bin/bash
# round 5, first GPU call: host facts, configs[1] at full size through the product (for the comparison with the reference binary's
# golden, tests/golden/configs1_full), kernel statistics of the CLI's default --sample regime, and the from-files baseline
# (ingest thread sweep) before the loader is rebuilt.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05a
rm -rf $o; mkdir -p $o
{ uname -r; nproc; lscpu | grep -E "Model name|Socket|Thread|Core|NUMA|L2|L3|Flags" | cut -c1-400; free -g | head -2; cat /sys/kernel/mm/transparent_hugepage/enabled; } > $o/host.txt 2>&1
timeout -k 10 420 python3 tests/fullsize_oracle_parity.py --against-golden gpurun_out/configs1_full > $o/configs1_full.txt 2>&1; echo "configs1 rc $?" >> $o/configs1_full.txt
tail -12 $o/configs1_full.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_default -- python3 tools/default_sample_leg.py > $o/default_sample_leg.txt 2>&1
cp $o/prof_default/*/*kernel_stats.csv $o/kernel_stats_default_sample.csv 2>/dev/null
python3 tools/kstats.py $o/prof_default 0.3 >> $o/default_sample_leg.txt 2>&1
tail -25 $o/default_sample_leg.txt
rm -rf $o/prof_default
E2E_THREAD_SWEEP=24,48,96 timeout -k 10 400 python3 tools/e2e_big.py 32000000 100 1 > $o/e2e_baseline.txt 2>&1
grep -v "^\[lhgt ingest\] staging" $o/e2e_baseline.txt | cut -c1-330 | tail -20
