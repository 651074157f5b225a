cd $GRAFT_REPO_ROOT
for a in 0 16 32 48; do
  PHASE_A_ONLY=1 PHASE_A_DEBUG=4194304 LHGT_PART_ABLATE=$a bash tools/r06/prof_kernels.sh r06b/abl_$a -- python3 $GRAFT_REPO_ROOT/tools/phase_a_time.py 100000000 > /dev/null
  echo "== LHGT_PART_ABLATE=$a"; grep part_ gpurun_out/r06b/abl_$a.kernels.txt
done
