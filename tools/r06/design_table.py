#!/usr/bin/env python3
"""the table of legs of DESIGN.md 5 from a bench_detail.json: design_table.py [detail.json]"""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r06/bench_driver_cmd_detail.json"))
rows = [("configs[2] (headline)", d)]
for n, s in d["secondary"].items():
    if isinstance(s, dict) and "value" in s:
        rows.append((n, s))
    elif isinstance(s, dict):
        for kk, v in s.items():
            if isinstance(v, dict) and "value" in v:
                rows.append((f"{n}/{kk}", v))
print("| leg | M pairs/s | ms per step | A | B | C | form of B | vote |\n|---|---|---|---|---|---|---|---|")
for n, s in rows:
    ph = s["phase_ms"]
    print(f"| `{n}` | {s['value']:.1f} | {s['ms_per_step']:.0f} | {ph['count_A']:.0f} | {ph['scan_B']:.0f} | {ph['vote_C']:.0f} | {s['scan_B_form']['form']} | {s['vote_form']['form']} |")
e = d.get("e2e", {})
b = e.get("batch_13g", {})
print("\nbatch:", {k: b.get(k) for k in ("value", "batch_s", "steady_sample_s", "steady_input_pairs_per_s_M", "break_even_samples")}, b.get("one_process_per_sample"))
print("samples:", [(x["total_s"], x["reference_s"], x["scan_s"], x["scan_form"]) for x in b.get("samples", [])])
print("e2e:", e.get("value"), {k: v.get("value") for k, v in (e.get("big") or {}).items() if isinstance(v, dict)})
print("slot_list_cost:", d.get("slot_list_cost"))
r = d["roofline"]
print("roofline:", {k: r.get(k) for k in ("kernel", "bound", "frac", "frac_raw", "frac_needed", "frac_of_bound", "launch_ms", "launches_per_step", "traffic")})
for ph, r in d["roofline_other"].items():
    print(ph, {k: r.get(k) for k in ("kernel", "bound", "frac", "frac_needed", "frac_of_bound")})
snp = d["secondary"].get("uhgg_deep_focused_snp1pct", {})
print("snp roofline:", {k: snp.get("roofline", {}).get(k) for k in ("kernel", "bound", "frac_needed", "bytes_needed", "needed_is")}, snp.get("work_stats"))
