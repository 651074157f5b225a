#!/bin/bash
# timing of rg_emit with stages removed (LHGT_RG_ABLATE; split and apply are skipped, the tables come out wrong): rocprofv3 kernel stats per variant
for a in 1 2 3 5; do
  export LHGT_RG_ABLATE=$a LEG_SCANS=2
  bash tools/r06/prof_kernels.sh r6_rg_ablate_$a -- python3 $GRAFT_REPO_ROOT/tools/r06/default_sample_leg.py | grep "rg_emit" | sed "s/^/ablate $a: /"
done
