#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gdb
timeout -k 10 420 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "set print thread-events off" -ex run -ex bt -ex "thread apply all bt 14" --args python3 tools/tmp/e2e_stress.py 32000000 40 > gpurun_out/gdb/stress_out.log 2> gpurun_out/gdb/stress_err.log
echo rc=$?
grep -v "New Thread\|exited\]\|Detaching" gpurun_out/gdb/stress_out.log | tail -150
tail -20 gpurun_out/gdb/stress_err.log
