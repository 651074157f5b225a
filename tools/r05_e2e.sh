#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05f; mkdir -p $o
E2E_KNOBS="${KNOBS:-two=;one=LHGT_ONE_COPY_STREAM:1;two_again=}" timeout -k 10 400 python3 tools/e2e_big.py 32000000 100 1 > $o/e2e.txt 2>&1
grep -v "staging + pinned" $o/e2e.txt | cut -c1-420 | tail -14
