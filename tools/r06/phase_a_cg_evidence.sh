#!/bin/bash
# Round 6, phase A: the key scatter's copy-out in whole 128-byte lines against round 4's 16-byte groups (LHGT_PART_CG=8) -- the A/B in one
# process, per-kernel times of both, and the bytes the L2 wrote and fetched (WRITE_SIZE / FETCH_SIZE, separate --pmc passes).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06b
mkdir -p $out
python3 tools/r06/phase_a_cg.py 100000000 4 > $out/phase_a_cg_ab.txt 2>&1
export PHASE_A_ONLY=1
LHGT_PART_CG=8 bash tools/r06/prof_kernels.sh r06b/ka_cg8 -- python3 $GRAFT_REPO_ROOT/tools/phase_a_time.py 100000000 > /dev/null
bash tools/r06/prof_kernels.sh r06b/ka_cg64 -- python3 $GRAFT_REPO_ROOT/tools/phase_a_time.py 100000000 > /dev/null
for c in WRITE_SIZE FETCH_SIZE; do
  LHGT_PART_CG=8 bash tools/r06/pmc_kernel.sh r06b/pmc_cg8 part_keys16_direct $c -- python3 $GRAFT_REPO_ROOT/tools/phase_a_time.py 100000000 > /dev/null
  bash tools/r06/pmc_kernel.sh r06b/pmc_cg64 part_keys16_direct $c -- python3 $GRAFT_REPO_ROOT/tools/phase_a_time.py 100000000 > /dev/null
done
{
  echo "== A/B in one process (tools/r06/phase_a_cg.py 100000000 4)"; cat $out/phase_a_cg_ab.txt
  echo "== kernels, LHGT_PART_CG=8 (16-byte groups)"; grep "part_" $out/ka_cg8.kernels.txt
  echo "== kernels, default (whole lines)"; grep "part_" $out/ka_cg64.kernels.txt
  echo "== L2 <-> memory bytes of part_keys16_direct (KiB; FETCH_SIZE counts 128-byte requests as 64), LHGT_PART_CG=8"; cat $out/pmc_cg8.txt
  echo "== the same, default"; cat $out/pmc_cg64.txt
} > $out/phase_a_whole_lines.txt
cat $out/phase_a_whole_lines.txt
