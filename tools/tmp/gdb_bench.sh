#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gdb
timeout -k 10 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "thread apply all bt 12" --args python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/gdb/out.log 2> gpurun_out/gdb/err.log
echo rc=$?
grep -v '^{' gpurun_out/gdb/out.log | grep -v "New Thread\|exited\]" | tail -120
tail -20 gpurun_out/gdb/err.log
