#!/bin/bash
# round 4: what vote_kernel_queued's fabric requests are: L2 hits / misses / fabric read requests and time of phase C on
# configs[2] (packed reference) per variant: plain loads vs non-temporal hints (debug bits 17 / 18), 4 MiB vs 2 MiB bitmap (LHGT_PF_BITS),
# and the stage ablations (bit 9: stop after the bitmap level; bit 10: before the peak_kmer gathers)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04h
rm -rf $o; mkdir -p $o
run() {  # tag, env PF bits, debug flags
  tag=$1; pf=$2; dbg=$3
  for c in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE TCP_TCC_READ_REQ_sum"; do
    LHGT_PF_BITS=$pf timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d /tmp/voteq_$tag -- python3 tools/vote_variants.py 100000000 $dbg > $o/$tag.txt 2>&1
    python3 - $tag /tmp/voteq_$tag >> $o/table.txt <<'PY'
import csv, glob, sys, collections
tag, d = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vote_kernel_queued" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in tot:
    print(f"{tag:28s} {c:24s} per_vote={tot[c] / (n[c] / 6):.4e}  (launches {n[c]})")
PY
    rm -rf /tmp/voteq_$tag
  done
  grep "debug" $o/$tag.txt | head -1 >> $o/table.txt
}
run plain_4MiB 25 393216
run nt_both_4MiB 25 0
run nt_both_2MiB 24 0
run stop_after_bitmap_4MiB 25 512
run stop_after_bitmap_2MiB 24 512
run no_peak_kmer_gathers_4MiB 25 1024
cat $o/table.txt
