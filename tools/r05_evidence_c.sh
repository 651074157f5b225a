#!/bin/bash
# round 5: loader evidence on one box -- (1) GPU tests that load from files (goldens, world2, fuzz), (2) from-files rates: --sample 1
# single pass vs the two planned passes, then the CLI's default --sample 2000000000 (plans + columns), (3) host ingest scaling
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05g; rm -rf $o; mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_world2.py tests/test_gpu_fuzz.py tests/test_gpu_refbinary.py -m gpu -x -q > $o/pytest_loader.txt 2>&1
tail -4 $o/pytest_loader.txt
E2E_KNOBS="single pass=;two planned passes (round 4's loader on this round's thread count)=LHGT_INGEST_STREAM:0;single pass again=" timeout -k 10 300 python3 tools/e2e_big.py 32000000 100 1 > $o/e2e_sample_1.txt 2>&1
grep -v "staging + pinned" $o/e2e_sample_1.txt | cut -c1-420 | tail -13
E2E_KNOBS="plans + columns=;plans + chunk loop=LHGT_INGEST_STREAM:0" timeout -k 10 300 python3 tools/e2e_big.py 32000000 100 2000000000 > $o/e2e_default_sample.txt 2>&1
grep -v "staging + pinned" $o/e2e_default_sample.txt | cut -c1-420 | tail -12
timeout -k 10 500 python3 tools/ingest_scaling.py 32000000 $o/ingest_scaling.txt > $o/ingest_scaling.log 2>&1
tail -34 $o/ingest_scaling.log | cut -c1-220
