#!/bin/bash
# round-6 evidence (late) on the final code: the driver's command (every line it prints + bench_detail.json), then the profiles of the headline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06f; mkdir -p $o
t0=$(date +%s)
timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver_cmd_all_lines.jsonl 2> $o/bench_driver_cmd.err || exit 1
echo "driver command: $(( $(date +%s) - t0 )) s"
grep '^{' $o/bench_driver_cmd_all_lines.jsonl | tail -1 > $o/bench_driver_cmd.json
cp bench_detail.json $o/bench_driver_cmd_detail.json
wc -c $o/bench_driver_cmd.json
tools/refresh_profiles.sh $o > $o/refresh.log 2>&1
tail -12 $o/refresh.log
