#!/usr/bin/env python3
"""print the kernels of a rocprofv3 --kernel-trace --stats run (directory) that take more than a given share"""
import csv, glob, sys
d, share = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("lhgt::", "").replace("void ", "")
    if float(r["Percentage"]) > share:
        print("%-30s calls=%4s avg_ms=%9.3f pct=%s" % (n, r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
