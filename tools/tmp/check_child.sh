#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py --e2e-child gpurun_out/e2e_child.json -k 32 -e 3 --device 0 > gpurun_out/e2e_child.log 2>&1; echo rc=$?
tail -c 400 gpurun_out/e2e_child.json; echo
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=5 > gpurun_out/t_parity.log 2>&1; echo rc=$?; tail -12 gpurun_out/t_parity.log
