#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 5 from an evidence directory (profiles/r02 or gpurun_out/<run>/profiles)."""
import csv
import json
import sys

d = sys.argv[1].rstrip("/")
b = json.load(open(f"{d}/bench_driver_cmd.json"))


def ph(p):
    return f"{p['count_A']:.0f} | {p['scan_B']:.0f} | {p['vote_C']:.0f}"


print("| workload | ms/step | M pairs/s | A | B | C | raw / filtered peaks |\n|---|---|---|---|---|---|---|")
print(f"| configs[2] 13 Gbase, 100 M pairs (default; driver command) | {b['ms_per_step']:.0f} | **{b['value']:.1f}** | {ph(b['phase_ms'])} | {b['raw_peaks']} / {b['filtered_peaks']} |")
s = b["secondary"]
for key, name in (("uhgg_packed_reference", "the same with the reference resident as packed bases (4.9 GB instead of 156 GB)"),
                  ("configs1_1g", "configs[1] 1 Gbase, 10 M pairs"), ("uhgg_focused_sample", "13 Gbase, 10 M pairs from 300 of its genomes (10x)"),
                  ("uhgg_default_sample", "13 Gbase under the default `--sample 2e9` (6.67 M pairs kept)")):
    v = s[key]
    rec = v["planted_transfers"]
    print(f"| {name} | {v['ms_per_step']:.0f} | {v['value']:.1f} | {ph(v['phase_ms'])} | {v['raw_peaks']} / {v['filtered_peaks']} (planted breakpoints inside an interval: {rec['inside_an_interval']} / {rec['breakpoints']}) |")
for kk, v in s.get("configs4_progenomes_1gpu", {}).items():
    if isinstance(v, dict):
        rec = v["planted_transfers"]
        print(f"| configs[4]-scale: 50 Gbase packed on ONE GPU, 25 M pairs from 300 genomes, {kk} | {v['ms_per_step']:.0f} | {v['value']:.1f} | {ph(v['phase_ms'])} | {v['raw_peaks']} / {v['filtered_peaks']} ({rec['inside_an_interval']} / {rec['breakpoints']}); B form {v['scan_B_form']['form']} |")
print()
print("| kernel (workload) | time / step | measured HBM bytes / step | achieved | frac of 8 TB/s | line fills or L2 requests per s (ceiling) |\n|---|---|---|---|---|---|")
r = b["roofline"]
rows = [("dominant: " + r["kernel"][:40], r)] + [(k, v) for k, v in b["roofline_other"].items()] + [("configs[1] " + s["configs1_1g"]["roofline"]["kernel"][:30], s["configs1_1g"]["roofline"])]
for name, v in rows:
    rr = v.get("request_rate") or {}
    print(f"| {name} | {v['ms_per_step']:.1f} ms ({v.get('launches_per_step')} launches) | {(v.get('traffic_per_step') or 0) / 1e12:.2f} TB | {(v.get('achieved') or 0) / 1e3:.2f} TB/s | **{v.get('frac')}** | {rr.get('value')} G/s ({rr.get('ceiling')}) |")
e = b["e2e"]
print("\ne2e:", json.dumps({k: e[k] for k in ("value", "total_s", "index_load_s", "reads_s", "kernels_ms")}), "build", e["with_index_build"], "packed", {k: e["with_packed_reference"][k] for k in ("value", "total_s", "reference_load_s", "same_peaks")})
print("cpu:", b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"], b["cpu_baseline"]["sample"][-50:])
print("compulsory:", b["compulsory"]["frac_of_peak"], "verify:", b["verify"]["ok"], "planted:", b["planted_transfers"])
for w in ("uhgg", "1g"):
    try:
        u = json.load(open(f"{d}/bench_{w}_under_rocprof.json"))
        rows = list(csv.DictReader(open(f"{d}/kernel_stats_{w}.csv")))
        dom = u["roofline"]["kernel"].split(" ")[0].split("(")[0]
        avg = [float(x["AverageNs"]) / 1e6 for x in rows if dom in x["Name"]]
        print(w, "bench launch_ms", u["roofline"]["launch_ms"], "rocprof avg of", dom, [round(a, 3) for a in avg[:2]])
        print("   ", [(x["Name"].split("(")[0].replace("lhgt::", "").replace("void ", "")[:28], x["Calls"], round(float(x["AverageNs"]) / 1e6, 2)) for x in rows[:12]])
    except Exception as ex:
        print(w, "n/a", ex)
