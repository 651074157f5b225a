#!/usr/bin/env python3
"""Summarise tools/pmc_collect.sh output: per kernel, counter totals over all its dispatches and per dispatch."""
import csv, glob, sys, collections, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("lhgt::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
rows = {}
for k in agg:
    d = {c: agg[k][c] / max(1, calls[k][c]) for c in agg[k]}   # per dispatch
    d["dispatches"] = max(calls[k].values())
    rows[k] = d
json.dump(rows, open(root + "/summary.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    if d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0) < 1000: continue
    print(f"{k[:28]:28s} n={d['dispatches']:4d} FETCH={d.get('FETCH_SIZE',0)/1e6:9.2f} GB(KiB-cnt)  WRITE={d.get('WRITE_SIZE',0)/1e6:9.2f}  "
          f"RDREQ={d.get('TCC_EA0_RDREQ_sum',0)/1e9:7.3f}G 32B={d.get('TCC_EA0_RDREQ_32B_sum',0)/1e9:7.3f}G BUB={d.get('TCC_BUBBLE_sum',0)/1e9:7.3f}G "
          f"HIT={d.get('TCC_HIT_sum',0)/1e9:7.3f}G MISS={d.get('TCC_MISS_sum',0)/1e9:7.3f}G")
