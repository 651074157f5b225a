#!/bin/bash
# round-6 evidence, part 2 (GPU box): the CLI's default regime by kernel -- the peak registry's direct kernel against the partition, rg_emit
# with stages removed -- and the dense vote with its bound off and on.  Output: gpurun_out/r06/
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06
mkdir -p $out
# the peaks' k-mers registered by partition: the default-sample leg by kernel, direct kernel against partition, and rg_emit with stages removed
LHGT_REGISTER_PART=0 LEG_SCANS=3 tools/r06/prof_kernels.sh r06/prof_reg_direct -- python3 $GRAFT_REPO_ROOT/tools/r06/default_sample_leg.py > /dev/null
( head -12 gpurun_out/r06/prof_reg_direct.kernels.txt; grep "^scan" gpurun_out/r06/prof_reg_direct.log ) > $out/kernel_stats_default_sample_direct_registry.txt
LEG_SCANS=3 tools/r06/prof_kernels.sh r06/prof_reg_part -- python3 $GRAFT_REPO_ROOT/tools/r06/default_sample_leg.py > /dev/null
( head -12 gpurun_out/r06/prof_reg_part.kernels.txt; grep "^scan" gpurun_out/r06/prof_reg_part.log ) > $out/kernel_stats_default_sample_registry_by_partition.txt
( echo "rg_emit per chunk (3 chunks of 2.56 G records; LHGT_RG_ABLATE, split and apply skipped): 1 no record is stored, 2 no pass 1, 3 the prologue alone";
  for a in 1 2 3; do LHGT_REGISTER_CHUNKS=3 LHGT_RG_ABLATE=$a LEG_SCANS=2 tools/r06/prof_kernels.sh r06/prof_reg_ab$a -- python3 $GRAFT_REPO_ROOT/tools/r06/default_sample_leg.py | grep "rg_emit" | sed "s/^/ablate $a: /"; done ) > $out/rg_emit_stage_ablation.txt
rm -rf gpurun_out/r06/prof_reg_*
# the dense vote's bound (k_vote.hip): off (debug bit 19) and on
( LHGT_DEBUG=$((1<<19)) LEG_SCANS=2 python3 tools/r06/default_sample_leg.py 2>&1 | grep "^scan" | sed "s/^/bound off: /"; LEG_SCANS=2 python3 tools/r06/default_sample_leg.py 2>&1 | grep "^scan" | sed "s/^/bound on:  /" ) > $out/dense_vote_bound_default_sample.txt
ls -la $out
