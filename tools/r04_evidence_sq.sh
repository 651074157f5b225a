#!/bin/bash
# round 4: SQ counters of phase A's kernels on one chunk of 8 Mi pairs -- round 3's sorted-tile scatters (debug bit 16) and the direct form
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04sq; rm -rf $o; mkdir -p $o
PHASE_A_DEBUG=65536 PHASE_A_ONLY=1 tools/sq_collect_cmd.sh $o/sq_phase_a_sorted_tiles.txt part_scatter_reads_reg,part_scatter_keys16,part_apply tools/phase_a_time.py 8388608 > /dev/null 2>&1
PHASE_A_ONLY=1 tools/sq_collect_cmd.sh $o/sq_phase_a_direct.txt part_reads_direct,part_keys16_direct,part_apply tools/phase_a_time.py 8388608 > /dev/null 2>&1
grep -c per_dispatch $o/*.txt
grep -E "INSTS_VALU|INSTS_LDS|LDS_BANK|WAIT_INST_LDS|WAVE_CYCLES" $o/*.txt
