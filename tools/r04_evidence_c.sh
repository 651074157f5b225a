#!/bin/bash
# round 4 evidence (GPU box): (1) the driver's bench command, its compact lines and its detail record; (2) rocprofv3 kernel stats of the headline
# and of the found-something workload (3 steps each, no extras) with the bench line of the same profiled run; (3) the PMC sums bench.py
# collected for both (--pmc-out).  Everything lands in gpurun_out/$1/profiles; copy what is to be judged into profiles/r04/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1; mkdir -p $out/profiles
p=$out/profiles
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $p/bench_driver_cmd_detail.json --pmc-out $p/pmc_live_uhgg.json > $out/bench_driver_cmd.log 2> $out/bench_driver_cmd.err || { echo "bench failed"; tail -5 $out/bench_driver_cmd.err; exit 1; }
grep '^{' $out/bench_driver_cmd.log > $p/bench_driver_cmd_all_lines.jsonl
tail -1 $p/bench_driver_cmd_all_lines.jsonl > $p/bench_driver_cmd.json
for wl in uhgg deep; do
  extra=""; [ $wl = deep ] && extra="--sample-contigs 300"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_$wl -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc --no-verify --no-stats --quiet $extra > $out/bench_${wl}_under_rocprof.log 2>&1
  grep '^{' $out/bench_${wl}_under_rocprof.log | tail -1 > $p/bench_${wl}_under_rocprof.json
  f=$(find $out/rocprof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $p/kernel_stats_$wl.csv
done
timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --sample-contigs 300 --no-cpu-baseline --no-extras --no-verify --quiet --pmc-out $p/pmc_live_deep.json > $out/bench_deep_run.log 2>&1
grep '^{' $out/bench_deep_run.log | tail -1 > $p/bench_deep_run.json
ls -la $p; tail -c 600 $p/bench_driver_cmd.json
