#!/usr/bin/env python3
"""profiles/traffic_per_launch.json from the PMC summaries bench.py writes with --pmc-out (tools/refresh_profiles.sh):
per workload tag, the HBM bytes and read requests per bench step of phase A's kernel family, ref_flags and the vote kernel,
stamped with the hash of the kernel sources they were measured on -- bench.py falls back to these figures only while that
stamp still matches (its own rocprofv3 --pmc passes come first).
usage: tools/make_traffic_json.py profiles/traffic_per_launch.json pmc_live_uhgg.json [pmc_live_1g.json ...]"""
import json
import os
import sys

dst, srcs = sys.argv[1], sys.argv[2:]
allw = json.load(open(dst)) if os.path.exists(dst) else {}
allw = {k: v for k, v in allw.items() if isinstance(v, dict) and "_stamp" in v}     # drop entries of the pre-stamp format
for src in srcs:
    d = json.load(open(src))
    ent = dict(d["per_step"])
    ent["_stamp"] = d["_stamp"]
    ent["_source"] = f"{src}: rocprofv3 --pmc passes of bench.py (1 step each), bytes = 2 x FETCH_SIZE + WRITE_SIZE (every read request is a 128-B line fill)"
    allw[d["tag"]] = ent
    print(d["tag"], {k: v for k, v in ent.items() if not k.startswith("_")})
json.dump(allw, open(dst, "w"), indent=1, sort_keys=True)
