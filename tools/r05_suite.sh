#!/bin/bash
# the GPU suite with per-test durations, then the from-files rate (3 runs) on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05e; mkdir -p $o
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=25 > $o/pytest_gpu.txt 2>&1
tail -40 $o/pytest_gpu.txt
E2E_THREAD_SWEEP=16 timeout -k 10 300 python3 tools/e2e_big.py 32000000 100 1 > $o/e2e.txt 2>&1
grep -v "staging + pinned" $o/e2e.txt | cut -c1-400 | tail -8
