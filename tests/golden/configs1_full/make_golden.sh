#!/bin/bash
# BASELINE configs[1] at FULL size through the REAL reference binary (oracle/_ref/extract_ref_z = the reference's own
# src/extract_ref_normal_peak.cpp compiled by oracle/build_ref.sh with the zero-new[] determinism shim):
#   1000 x 1 Mbp reference, 10 M 150 bp pairs, k = 32, e = 3, seed 1, --sample 1, max_peak 300000000 (the CLI's default)
#   -t 1, and -t 10 with its threads run in creation order (oracle/_ref/libseqthreads.so: the -t N contract).
# The inputs come from tests/synth_cpu.c, the host twin of the device generator the GPU box uses (same bytes: inputs.sha256).
# About 3 hours of ONE host core and 23 GB of memory; run once per round in the build container, in the background:
#   tests/golden/configs1_full/make_golden.sh /tmp/c1full
# Outputs copied next to this script: interval_t1.txt, interval_t10.txt, genome.len.txt.sha256, inputs.sha256, meta.txt.
# tests/fullsize_oracle_parity.py --against-golden compares the product with them on the GPU box.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../../.." && pwd)"
W="${1:-/tmp/c1full}"
REF="$root/oracle/_ref"
[ -x "$REF/extract_ref_z" ] || { echo "oracle/_ref/extract_ref_z missing: run oracle/build_ref.sh where /root/reference exists" >&2; exit 1; }
mkdir -p "$W"
gcc -O2 -fopenmp -o "$W/synth_cpu" "$root/tests/synth_cpu.c"
if [ ! -f "$W/s.2.fq" ]; then "$W/synth_cpu" "$W" 1000 1000000 10000000; fi
(cd "$W" && sha256sum ref.fa s.1.fq s.2.fq > inputs.sha256 && cat inputs.sha256)
ARGS="s.1.fq s.2.fq ref.fa"
TAIL="0.1 0.08 %T 32 300000000 3 1 1"
cd "$W"
t0=$(date +%s)
"$REF/extract_ref_z" $ARGS interval_t1.txt ${TAIL/\%T/1} > log_t1.txt 2>&1
t1=$(date +%s)
echo "-t 1 done in $((t1 - t0)) s: $(wc -l < interval_t1.txt) interval lines; $(grep -h 'raw BKPs' log_t1.txt | tail -1)"
LD_PRELOAD="$REF/libseqthreads.so" "$REF/extract_ref_z" $ARGS interval_t10.txt ${TAIL/\%T/10} > log_t10.txt 2>&1
t2=$(date +%s)
echo "-t 10 (threads in creation order) done in $((t2 - t1)) s: $(wc -l < interval_t10.txt) interval lines"
# the index file without bytes 1198-1199 (the reference writes two bytes from behind its coder array there)
python3 - > outputs.sha256 <<'PY'
import hashlib
h = hashlib.sha256(open("ref.fa.genome.len.txt", "rb").read()).hexdigest()
print(f"{h}  ref.fa.genome.len.txt")
g = hashlib.sha256()
with open("ref.fa.k32.h3.index.dat", "rb") as f:
    head = bytearray(f.read(1200)); head[1198:1200] = b"\0\0"; g.update(bytes(head))
    for blk in iter(lambda: f.read(1 << 24), b""):
        g.update(blk)
print(f"{g.hexdigest()}  ref.fa.k32.h3.index.dat (bytes 1198-1199 zeroed)")
PY
cp interval_t1.txt interval_t10.txt inputs.sha256 outputs.sha256 "$here/"
{
  echo "made by tests/golden/configs1_full/make_golden.sh on $(date -u +%Y-%m-%d) with oracle/_ref/extract_ref_z ($(sha256sum "$REF/extract_ref_z" | cut -c1-16))"
  echo "-t 1: $((t1 - t0)) s (index built in that run); -t 10 sequential threads: $((t2 - t1)) s"
  grep -h 'raw BKPs\|Finish with time\|K-mer counting' log_t1.txt | sed 's/^/t1: /'
  grep -h 'raw BKPs\|Finish with time\|K-mer counting' log_t10.txt | sed 's/^/t10: /'
} > "$here/meta.txt"
cat "$here/meta.txt"
