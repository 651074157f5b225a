#!/bin/bash
# round-2 evidence, part B (GPU box): the -m gpu suite, the RCCL exchange code at world size 1 (replicated and reference-sharded
# phase B), the packed-reference bench line, the from-FASTQ trace
out=gpurun_out/$1; mkdir -p $out/profiles
timeout -k 10 700 python -m pytest tests -m gpu -x -q --durations=8 > $out/pytest.log 2>&1 || { echo "pytest failed"; tail -20 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist --no-extras --no-cpu-baseline 2> /dev/null | grep '^{' | tail -1 > $out/profiles/bench_force_dist.json || exit 1
timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist --shard-index --no-extras --no-cpu-baseline 2> /dev/null | grep '^{' | tail -1 > $out/profiles/bench_force_dist_sharded.json || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 5 --warmup 2 --ref-form packed --no-extras --no-cpu-baseline --pmc-out $out/profiles/pmc_live_uhgg_packed.json 2> /dev/null | grep '^{' | tail -1 > $out/profiles/bench_uhgg_packed_run.json || exit 1
LHGT_INGEST_TRACE=1 timeout -k 10 200 python tools/e2e_files.py > $out/profiles/e2e_trace.txt 2>&1 || exit 1
LHGT_REF_FORM=packed LHGT_INGEST_TRACE=1 timeout -k 10 200 python tools/e2e_files.py 4000000 1000 > $out/profiles/e2e_trace_packed_1gbase.txt 2>&1 || exit 1
ls -la $out/profiles
