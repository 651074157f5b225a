#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_refbinary.py -x -q -m gpu -k "cpu_restatement" > gpurun_out/t_sl1.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_sl1.log
LHGT_TRACE=1 timeout -k 10 400 python3 tools/ragged_vote_stages.py > gpurun_out/rvs.log 2>&1; echo rc=$?; grep -v "amdgpu.ids\|lhgt\] table\|tiles\|peak_kmer:" gpurun_out/rvs.log | cut -c1-300
