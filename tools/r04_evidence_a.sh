#!/bin/bash
# round 4: phase A's direct form -- stage ablation (LHGT_PART_ABLATE) per geometry (LHGT_PART_GEOM), then SQ counters of its kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04c
rm -rf $o; mkdir -p $o
for geom in ${GEOMS:-0 1}; do
for ab in ${ABLATES:-0 1 2 3 16 32 48}; do
  PHASE_A_DEBUG=$(( (geom == 0 ? 1 << 21 : 0) | 1 << 22 )) LHGT_PART_ABLATE=$ab PHASE_A_ONLY=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $o/g${geom}ab$ab -- python3 tools/phase_a_time.py 25000000 > $o/g${geom}ab$ab.txt 2>&1
done
done
python3 - <<'PY' > gpurun_out/r04c/ablation.txt
import sqlite3, glob, re
for d in sorted(glob.glob("gpurun_out/r04c/g*ab*/")):
    for f in glob.glob(d + "*/*_results.db"):
        db = sqlite3.connect(f); cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
        q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%part_%' group by s.kernel_name"
        print(d.split('/')[-2], "  ".join(f"{re.sub(r'^_ZN4lhgt[0-9]+', '', r[0])[:18]} n={r[1]} {r[2]:.2f} ms" for r in cur.execute(q)))
PY
cat gpurun_out/r04c/ablation.txt
if [ -n "$SQ" ]; then
PHASE_A_DEBUG=$(( SQ == 0 ? 1 << 21 : 0 )) PHASE_A_ONLY=1 tools/sq_collect_cmd.sh $o/sq_phase_a_direct.txt part_reads_direct,part_keys16_direct,part_apply tools/phase_a_time.py 8000000 > /dev/null 2>&1
grep -E "INSTS_VALU|INSTS_SALU|INSTS_LDS|WAVE_CYCLES|ACTIVE_INST_ANY|ACTIVE_INST_VALU|WAIT_ANY|WAIT_INST_ANY|LDS_BANK|LDS_IDX|WAIT_INST_LDS" $o/sq_phase_a_direct.txt
fi
