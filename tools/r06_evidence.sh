#!/bin/bash
# round-6 evidence (GPU box): the SNP 1 % leg dense vs shared (times, digests, line fills), its kernels under rocprofv3, the stage
# ablation of vs_probe, fabric read requests per pair before / after (TCC_EA0_RDREQ), SQ counters of vs_probe.  Output: gpurun_out/r06/
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06
mkdir -p $out
LHGT_TRACE=1 python3 tools/r06/snp_leg.py 100000000 300 2>&1 | grep -v "amdgpu.ids\|tiles\|table:" > $out/snp_leg_dense_vs_shared.txt
python3 tools/r06/vs_ablate.py 100000000 300 2>&1 | grep -v amdgpu.ids > $out/vs_probe_stage_ablation.txt
tools/r06/prof_kernels.sh r06/prof_snp -- python3 $GRAFT_REPO_ROOT/tools/r06/snp_leg.py 100000000 300 > /dev/null
cp gpurun_out/r06/prof_snp.kernels.txt $out/kernel_stats_snp_leg.txt
cp gpurun_out/r06/prof_snp/k_kernel_stats.csv $out/kernel_stats_snp_leg.csv
rm -rf gpurun_out/r06/prof_snp gpurun_out/r06/prof_snp.log
# fabric read requests: the dense vote (debug bit 28) against the shared form, 30 M pairs (one vote each)

LHGT_DEBUG=$((1<<28)) tools/r06/pmc_kernel.sh r06/pmc_dense vote_kernel "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" -- python3 $GRAFT_REPO_ROOT/tools/r06/snp_shared_only.py 30000000 90 1 > $out/rdreq_dense_vote_kernel.txt
tools/r06/pmc_kernel.sh r06/pmc_shared vs_probe "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" -- python3 $GRAFT_REPO_ROOT/tools/r06/snp_shared_only.py 30000000 90 1 > $out/rdreq_shared_vs_probe.txt
tools/r06/pmc_kernel.sh r06/pmc_shared2 vote_kernel "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" -- python3 $GRAFT_REPO_ROOT/tools/r06/snp_shared_only.py 30000000 90 1 > $out/rdreq_shared_generic_rest.txt
tools/r06/pmc_kernel.sh r06/pmc_sq1 vs_probe "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 $GRAFT_REPO_ROOT/tools/r06/snp_shared_only.py 30000000 90 1 > $out/sq_vs_probe_1.txt
tools/r06/pmc_kernel.sh r06/pmc_sq2 vs_probe "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python3 $GRAFT_REPO_ROOT/tools/r06/snp_shared_only.py 30000000 90 1 > $out/sq_vs_probe_2.txt
rm -rf gpurun_out/r06/pmc_* 
ls -la $out
# the shared form at lower coverage: 10 M pairs from 300 genomes (5.6 reads per bucket), 50 M pairs with SNPs (23.8)
( LHGT_TRACE=1 python3 tools/r06/snp_leg.py 10000000 300 0 2>&1 | grep "form\|can vote\|keyed"; LHGT_TRACE=1 python3 tools/r06/snp_leg.py 50000000 300 10 2>&1 | grep "form\|can vote\|keyed" ) > $out/shared_vote_at_lower_coverage.txt
ls -la $out
bash tools/r06/registry_evidence.sh
