"""The ONE line the driver parses: short enough to survive any tail buffer (< 4 KB), printed after every stage of the run so that
the last complete line on stdout always carries the timed headline.  Everything else goes to bench_detail.json."""
import json

LIMIT = 4096

ROOF_KEYS = ("kernel", "bound", "peak", "unit", "launch_ms", "launches_per_step", "traffic", "achieved", "frac", "frac_raw", "frac_model",
             "model_exceeded", "frac_needed", "overfetch", "l2_hit_rate", "frac_of_bound")


def _roof(r):
    if not isinstance(r, dict):
        return None
    out = {k: r[k] for k in ROOF_KEYS if k in r}
    rr = r.get("request_rate")
    if isinstance(rr, dict):
        out["request_rate"] = {"value": rr.get("value"), "ceiling": rr.get("ceiling"), "unit": "G req/s"}
    bc = r.get("bound_ceiling")
    if isinstance(bc, dict) and "value" in bc:              # what bounds the kernel: achieved / ceiling in the bound's own unit (frac_of_bound)
        out["bound_ceiling"] = {"value": bc.get("value"), "ceiling": bc.get("ceiling"), "unit": bc.get("unit")}
    return out


def _roof_short(r):
    """a secondary roofline entry: what ran, what bounds it, how close"""
    if not isinstance(r, dict):
        return None
    return {k: r[k] for k in ("kernel", "bound", "launch_ms", "launches_per_step", "frac", "frac_needed", "frac_of_bound") if r.get(k) is not None}


def _num(x, nd=3):
    return round(x, nd) if isinstance(x, float) else x


def _leg(d):
    """one secondary leg: its value, what it found, its dominant kernel's fraction"""
    if not isinstance(d, dict) or "value" not in d:
        return None
    out = {"value": _num(d["value"])}
    pt = d.get("planted_transfers")
    if isinstance(pt, dict):
        out["recall"] = pt.get("recall")
    if "filtered_peaks" in d:
        out["filtered_peaks"] = d["filtered_peaks"]
    r = d.get("roofline")
    if isinstance(r, dict) and r.get("frac") is not None:
        out["frac"] = r["frac"]
    return out


def compact_line(detail):
    """detail = the full record bench.py keeps (and writes to bench_detail.json); returns the dict to print"""
    keep = ("metric", "value", "unit", "n_gpus", "world_size", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "backend", "exchange_ms", "n1_equivalent_ms", "raw_peaks", "filtered_peaks", "stage")
    line = {k: detail[k] for k in keep if k in detail and detail[k] is not None or k in ("vs_baseline",)}
    cfg = detail.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "pairs_per_gpu", "ref_bases", "k", "e", "parallelism") if k in cfg}
    if "phase_ms" in detail:
        line["phase_ms"] = {k: _num(v, 1) for k, v in detail["phase_ms"].items()}
    pt = detail.get("planted_transfers")
    if isinstance(pt, dict):
        line["planted_transfers"] = {"recall": pt.get("recall"), "breakpoints": pt.get("breakpoints")}
    line["roofline"] = _roof(detail.get("roofline"))
    rb = (detail.get("roofline_other") or {}).get("ref_flags")
    if isinstance(rb, dict):                                # the k-mer-scan kernel of the reference side (north_star's HBM target), when it is not the dominant one
        line["scan_roofline"] = _roof_short(rb)
    found = (detail.get("secondary") or {}).get("uhgg_deep_focused_sample")
    if isinstance(found, dict) and "value" in found:      # the same read count where the path finds its planted transfers
        line["value_found"] = _num(found["value"])
        line["found"] = {"workload": found.get("workload_short", "13 Gbase ref, 100 M pairs from 300 of its contigs (100x)"),
                         "ms_per_step": found.get("ms_per_step"), "phase_ms": {k: _num(v, 1) for k, v in found.get("phase_ms", {}).items()},
                         "recall": (found.get("planted_transfers") or {}).get("recall"), "filtered_peaks": found.get("filtered_peaks"),
                         "roofline": _roof_short(found.get("roofline"))}
    sl = detail.get("slot_list_cost")
    if isinstance(sl, dict):                              # the headline runs on a per-reference precompute: what it costs, when it has paid for itself
        for kk in ("slot_list_build_ms", "value_without_list", "break_even_samples"):
            if sl.get(kk) is not None:
                line[kk] = sl[kk]
    ix = (detail.get("secondary") or {}).get("uhgg_index_form")
    if isinstance(ix, dict) and "value" in ix:            # the same workload with the index file's hashes resident: what rounds 1-4 led with
        line["value_index_form"] = _num(ix["value"])
    fx = (detail.get("secondary") or {}).get("uhgg_deep_focused_index_form")
    if isinstance(fx, dict) and "value" in fx:
        line["value_found_index_form"] = _num(fx["value"])
    sh = detail.get("sharded_index")
    if isinstance(sh, dict):
        line["sharded_index"] = {"value": sh.get("value"), "ms_per_step": sh.get("ms_per_step"), "same_peaks": sh.get("same_peaks")}
    cb = detail.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: cb[k] for k in ("value", "unit", "cores", "kind", "identical_to_gpu", "error") if k in cb}
        if "sample" in cb:                                  # cut at a word, and say that it was cut
            smp = cb["sample"]
            c["sample"] = smp if len(smp) <= 110 else smp[:107].rsplit(" ", 1)[0] + " ..."
        ref = cb.get("reference")
        if isinstance(ref, dict):
            c["reference"] = {k: ref[k] for k in ("value", "threads", "wall_s", "own_clock_s", "identical_to_gpu", "error") if k in ref}
        line["cpu_baseline"] = c
    sec = {}
    for name, d in (detail.get("secondary") or {}).items():
        if name == "uhgg_deep_focused_sample":
            continue
        if name in ("configs4_progenomes_1gpu", "configs4_as_named") and isinstance(d, dict):
            for kk in ("k32", "k21"):
                if _leg(d.get(kk)):
                    tag = f"configs4_{kk}" if name == "configs4_progenomes_1gpu" else f"configs4_named_{kk}"
                    sec[tag] = _leg(d[kk])
                    if "input_pairs_per_s_M" in d[kk]:
                        sec[tag]["input"] = d[kk]["input_pairs_per_s_M"]
        elif _leg(d):
            sec[name] = _leg(d)
        elif isinstance(d, str):
            sec[name] = d[:80]
    e2e = detail.get("e2e")
    if isinstance(e2e, dict):
        if "value" in e2e:
            sec["e2e_4m_pairs"] = {"value": e2e["value"]}
        for tag, d in (e2e.get("big") or {}).items():
            if isinstance(d, dict) and "value" in d:
                sec[f"e2e_32m_{tag}"] = {"value": d["value"]}
        bt = e2e.get("batch_13g")
        if isinstance(bt, dict) and "value" in bt:        # extract_ref --batch from files at 13 Gbase, reference load and list build included
            sec["batch_13g_from_files"] = {"value": bt["value"], "steady": bt.get("steady_input_pairs_per_s_M"), "break_even_samples": bt.get("break_even_samples")}
        elif isinstance(bt, dict):
            sec["batch_13g_from_files"] = str(bt.get("error") or bt.get("skipped"))[:80]
        if "error" in e2e:
            sec["e2e_error"] = str(e2e["error"])[:80]
    if sec:
        line["secondary"] = sec
    line["detail"] = "bench_detail.json"
    # never longer than the limit: shed the least needed parts first
    for drop in (("secondary",), ("found", "roofline"), ("cpu_baseline", "sample"), ("found",), ("cpu_baseline", "reference")):
        if len(json.dumps(line)) < LIMIT - 64:
            break
        tgt = line
        for kk in drop[:-1]:
            tgt = tgt.get(kk, {}) if isinstance(tgt.get(kk), dict) else {}
        tgt.pop(drop[-1], None)
    return line


def emit(detail, stage, detail_path=None):
    """print the compact line (stage = how far the run has come) and refresh the detail file"""
    import ctypes
    import sys
    detail["stage"] = stage
    if detail_path:
        try:
            with open(detail_path + ".tmp", "w") as f:
                json.dump(detail, f, indent=1, default=str)
            import os
            os.replace(detail_path + ".tmp", detail_path)
        except OSError:
            pass
    # RCCL writes its version banner through C stdio, which on a pipe is flushed at exit, i.e. after Python's output:
    # flush it now so that a JSON line is the last thing on stdout
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    s = json.dumps(compact_line(detail))
    assert len(s) < LIMIT, len(s)
    print(s, flush=True)
    return s
