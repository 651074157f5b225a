#!/bin/bash
# round-3 evidence, part B (GPU box): the exchange code over RCCL at world size 1 (replicated and reference-sharded phase B), the
# N = 2 path of bench.py with both ranks on the one GPU (gloo, host-staged), phase times against the CU share, load policies on
# L2-resident tables, the Infinity-Cache share estimate, extract_ref from 20 GB of FASTQ
out=gpurun_out/$1; mkdir -p $out/profiles
timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist --no-extras --no-cpu-baseline 2> /dev/null | grep '^{' | tail -1 > $out/profiles/bench_force_dist.json || exit 1
timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist --shard-index --no-extras --no-cpu-baseline 2> /dev/null | grep '^{' | tail -1 > $out/profiles/bench_force_dist_sharded.json || exit 1
timeout -k 10 200 python3 bench.py --gpus 2 --workload 1g --backend gloo --steps 3 --warmup 1 2> /dev/null | grep '^{' | tail -1 > $out/profiles/bench_n2_one_gpu_gloo_1g.json || exit 1
timeout -k 10 300 python3 tools/cu_share.py > $out/profiles/cu_share_phase_times.txt 2>&1 || exit 1
timeout -k 10 100 ./tools/probe_policy > $out/profiles/probe_policy_l2_resident.txt 2>&1 || exit 1
timeout -k 10 100 ./tools/probe_shapes mall > $out/profiles/probe_mall_share.txt 2>&1 || exit 1
ls -la $out/profiles
