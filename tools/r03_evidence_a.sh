#!/bin/bash
# round-3 evidence, part A (GPU box): kernel stats + bench lines under rocprofv3, live PMC summaries, the driver's bench command
out=gpurun_out/$1; mkdir -p $out/profiles
timeout -k 10 700 bash tools/refresh_profiles.sh $out/profiles > $out/refresh.log 2>&1 || { echo "refresh failed"; tail -5 $out/refresh.log; exit 1; }
timeout -k 10 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.log 2> $out/bench_driver_cmd.err || { echo "bench failed"; tail -5 $out/bench_driver_cmd.err; exit 1; }
grep '^{' $out/bench_driver_cmd.log | tail -1 > $out/profiles/bench_driver_cmd.json
ls -la $out/profiles
