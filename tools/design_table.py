#!/usr/bin/env python3
"""the table of the driver's command and the sentence on its roofline entries, rewritten from
profiles/r05/bench_driver_cmd_detail.json (so that the document follows the committed evidence, not a memory of it)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dd = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_driver_cmd_detail.json")))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
rows = [("configs[2] (headline)", dd["value"], dd["ms_per_step"], dd["phase_ms"], dd["scan_B_form"]["form"], dd["vote_form"]["form"])]
for n, x in dd["secondary"].items():
    if isinstance(x, dict) and "phase_ms" in x:
        rows.append((n, x["value"], x["ms_per_step"], x["phase_ms"], x["scan_B_form"]["form"], x["vote_form"]["form"]))
    elif isinstance(x, dict):
        for kk, y in x.items():
            if isinstance(y, dict) and "phase_ms" in y:
                rows.append((n + "/" + kk, y["value"], y["ms_per_step"], y["phase_ms"], y["scan_B_form"]["form"], y["vote_form"]["form"]))
out = ["| leg | M pairs/s | ms per step | A | B | C | form of B | vote |", "|---|---|---|---|---|---|---|---|"]
for n, v, ms, ph, sf, vf in rows:
    out.append(f"| `{n}` | {v:.1f} | {ms:.0f} | {ph['count_A']:.0f} | {ph['scan_B']:.0f} | {ph['vote_C']:.0f} | {sf} | {vf} |")
a = s.index("| leg | M pairs/s | ms per step | A | B | C | form of B | vote |")
b = s.index("\nThe roofline entries of the headline:")
s = s[:a] + "\n".join(out) + "\n" + s[b:]
r, rb, ra = dd["roofline"], dd["roofline_other"]["ref_flags"], dd["roofline_other"]["count_A"]
e2e = dd["e2e"]["big"]
a = s.index("The roofline entries of the headline:")
b = s.index("**Round 5: what bounds each kernel, and the figures from files.**")
new = (f"The roofline entries of the headline: phase C `vote_kernel_queued` `bound: l2_requests`, `frac_of_bound` {r['frac_of_bound']} "
       f"({r['bound_ceiling']['value']} of 269.5 G requests/s), `frac` {r['frac']} of the HBM peak; phase B `no_kmer_flags+ref_single_slots+ref_trio_runs` "
       f"`bound: hbm_lines`, `frac_of_bound` {rb.get('frac_of_bound')} ({rb['bound_ceiling'].get('value')} of 56 G line fills/s), `frac` {rb.get('frac')}, "
       f"`frac_needed` {rb.get('frac_needed')}; phase A `lds_random` {ra.get('frac_of_bound')}.  `uhgg_default_sample` (the synthetic half-of-the-catalogue "
       "sample under the CLI's down-sampling) stays where it was: its phase B is `register_peaks`' 7.7 G atomics.  From files on this box "
       f"**{e2e['sample_1']['value']} / {e2e['default_sample_2e9']['value']} M input pairs/s** (`e2e_32m_*`; 56-102 by box over the round's runs: see "
       "\"Which socket\", §4).\n\n")
s = s[:a] + new + s[b:]
open(p, "w").write(s)
print("\n".join(out))
