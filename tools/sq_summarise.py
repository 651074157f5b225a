#!/usr/bin/env python3
"""Per-kernel totals of SQ counters from a rocprofv3 --pmc run (instruction mix, wait shares)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("lhgt::", "").replace("void ", "")
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print("%-28s n=%5d VALU=%.2e SALU=%.2e LDS=%.2e wavecyc=%.2e active=%.2e wait_any=%.2e wait_inst=%.2e busy=%.2e" % (
        k[:28], n[k], v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"], v["SQ_INSTS_LDS"], v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_ANY"],
        v["SQ_WAIT_ANY"], v["SQ_WAIT_INST_ANY"], v["SQ_BUSY_CYCLES"]))
