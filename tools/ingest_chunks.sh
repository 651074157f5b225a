#!/bin/bash
# chunk size of the single-pass loader: does a column that fits a core's L2 parse faster? host-only probe, then the whole call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05d; mkdir -p $o
d=/tmp/ing; rm -rf $d; mkdir -p $d
python3 - <<'PY'
import sys, time
sys.path.insert(0, ".")
import bench
bench.synth_files_sliced("/tmp/ing", 32, 3, 100, 1_000_000, 32000000, 0)
PY
: > $o/chunks.txt
for ch in 2097152 1048576 524288 262144; do
  for rep in 1 2; do
    echo "== host-only chunk=$ch" >> $o/chunks.txt
    LHGT_INGEST_CHUNK_BYTES=$ch LHGT_INGEST_TRACE=1 python3 tools/ingest_scaling.py --worker single $d/s.1.fq $d/s.2.fq 0 1 16 1 /tmp >> $o/chunks.txt 2>&1
  done
done
python3 - >> $o/chunks.txt 2>&1 <<'PY'
import os, sys, json, time
sys.path.insert(0, ".")
os.environ["LHGT_INGEST_TRACE"] = "1"
from localhgt_amd import extract_ref
d = "/tmp/ing"
def run(tag):
    a = extract_ref.Args(d + "/s.1.fq", d + "/s.2.fq", d + "/ref.fa", d + "/interval.txt", 0.1, 0.08, 10, 32, 300_000_000, 3, 1, 1.0)
    rep = extract_ref.run(a, log=lambda *x: None)
    print(tag, json.dumps({k: round(v, 3) if isinstance(v, float) else v for k, v in rep.items() if k in ("pairs_kept", "reads_s", "total_s", "count_kernel_ms", "n_filtered")}),
          f"-> {rep['pairs_seen'] / rep['total_s'] / 1e6:.1f} M input pairs/s", flush=True)
run("index built in-run")
for ch in (2097152, 1048576, 524288, 262144, 2097152):
    os.environ["LHGT_INGEST_CHUNK_BYTES"] = str(ch)
    for i in range(3):
        run(f"e2e chunk={ch}")
PY
grep -E "^==|one pass|^e2e|^index" $o/chunks.txt | sed -e 's/columns of [0-9]* + [0-9]* bytes, //' -e 's/of \/tmp\/ing\/s.1.fq//' | cut -c1-300
