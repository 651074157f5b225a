#!/bin/bash
# round-2 evidence run on the GPU box: full -m gpu suite, profiles (kernel stats + PMC + bench lines), the driver's bench command,
# the RCCL path at world size 1, the from-FASTQ trace
out=gpurun_out/$1; mkdir -p $out
python -m pytest tests -m gpu -x -q --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" > $out/rc.txt
tail -14 $out/pytest.log
bash tools/refresh_profiles.sh $out/profiles > $out/refresh.log 2>&1; echo "refresh rc=$?" >> $out/rc.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.log 2>&1; echo "bench rc=$?" >> $out/rc.txt
grep '^{' $out/bench_driver_cmd.log | tail -1 > $out/bench_driver_cmd.json
python3 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist --no-extras --no-cpu-baseline > $out/bench_force_dist.log 2>&1; echo "force-dist rc=$?" >> $out/rc.txt
python3 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist --shard-index --no-extras --no-cpu-baseline > $out/bench_force_dist_sharded.log 2>&1; echo "force-dist sharded rc=$?" >> $out/rc.txt
LHGT_INGEST_TRACE=1 python tools/e2e_files.py > $out/e2e_trace.txt 2>&1
cat $out/rc.txt; tail -3 $out/e2e_trace.txt
