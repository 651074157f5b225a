"""HBM traffic of the kernels of one bench step, measured live: child runs of bench.py (1 step, no extras) under
`rocprofv3 --pmc`, one child per counter group (counters only: no trace domain besides the dispatch records the CSV needs).
The children must run while this process holds nothing on the GPU: the headline workload fills most of the 288 GB."""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

from . import ROOT

# FETCH_SIZE tallies 64 B per fabric read request (TCC_EA0_RDREQ), but on gfx950 EVERY request of these kernels is a 128-B line
# fill: the guide says so for wide streaming reads, and tools/probe_shapes.hip calibrates it for random 4-byte probes (two probes
# in the two 64-B halves of one line cost ONE request: split128, 1.07 requests per pair; a 16 B/lane stream shows 128.0 B per
# request; TCC_BUBBLE and the 32-B request counter are zero) -- profiles/r02/probe_shapes_pmc.txt.  So read bytes = FETCH_SIZE x 2.
FETCH_SIZE_SCALE = 2
PMC_PASSES = [["FETCH_SIZE", "TCP_TCC_READ_REQ_sum"], ["WRITE_SIZE", "TCC_EA0_RDREQ_sum"]]
PMC_PASS_L2 = ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"]            # L2 hit rate (MI355X_MICROARCH.md: TCC_HIT / (TCC_HIT + TCC_MISS))
KERNEL_SOURCES = {   # which sources a kernel's measured traffic depends on (stamp of profiles/traffic_per_launch.json)
    "count_A": ("k_count_part.hip", "k_count.hip"), "ref_flags": ("k_scan.hip",), "vote_kernel": ("k_vote.hip",),
}
COMMON_SOURCES = ("lhgt_hash.hpp", "lhgt_common.hpp", "k_ingest.hip", "k_synth.hip")
PHASE_KERNELS = {
    "count_A": ("part_scatter_reads", "part_scatter_keys", "part_reads_direct", "part_keys16_direct", "part_apply", "count_direct"),
    "ref_flags": ("ref_flags=", "ref_flags_lite=", "ref_flags_trio=",       # "=": the whole name (ref_flags_fill belongs to the few unsettled tiles)
                  "ref_flags_slots=", "no_kmer_flags=", "contig_tail_flags=", "ref_single_slots=", "ref_trio_runs="),   # round 5: the list forms' kernels
    "vote_kernel": ("vote_kernel",),
}
WORKLOAD_FLAGS = ("workload", "pairs", "contigs", "contig_len", "k", "e", "count_mode", "debug", "sample_contigs", "ref_form", "snp")


def _kernel_in(kname, names):
    return any(kname == n[:-1] if n.endswith("=") else kname.startswith(n) for n in names)


def source_stamp(names):
    h = hashlib.sha256()
    for n in sorted(set(names) | set(COMMON_SOURCES)):
        with open(os.path.join(ROOT, "localhgt_amd", "csrc", n), "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()[:16]


def child_command(args):
    """bench.py on the same workload: one step, nothing but the timed path"""
    py = sys.executable if os.path.basename(sys.executable).startswith("python") else "python3"
    return [py, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--no-pmc", "--no-verify", "--no-stats",
            "--quiet", "--workload", args.workload, "--pairs", str(args.pairs), "--contigs", str(args.contigs),
            "--contig-len", str(args.contig_len), "-k", str(args.k), "-e", str(args.e), "--count-mode", str(args.count_mode),
            "--debug", str(args.debug), "--sample-contigs", str(args.sample_contigs), "--snp", str(args.snp),
            "--ref-form", args.ref_form] + (["--ragged"] if args.ragged else []) + (["--no-slot-list"] if getattr(args, "no_slot_list", False) else [])


def collect_pmc(args, passes, timeout_s=420):
    """{kernel base name: {counter: sum over the step's dispatches, 'dispatches': n}}, notes"""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    agg, notes = {}, []
    base = child_command(args)
    for counters in passes:
        d = tempfile.mkdtemp(prefix="lhgt_pmc_", dir="/tmp")
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                     "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            env["TMPDIR"] = "/tmp"
            res = subprocess.run([exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--"] + base, cwd="/tmp", env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not files:
                notes.append(f"pass {'+'.join(counters)}: rc {res.returncode}, {len(files)} csv")
                continue
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        kname = r["Kernel_Name"].replace("void ", "").replace("lhgt::", "").split("(")[0].split("<")[0]
                        ent = agg.setdefault(kname, {"_n": {}})
                        c = r["Counter_Name"]
                        ent[c] = ent.get(c, 0.0) + float(r["Counter_Value"])
                        ent["_n"][c] = ent["_n"].get(c, 0) + 1
        except subprocess.TimeoutExpired:
            notes.append(f"pass {'+'.join(counters)}: timeout")
        except Exception as ex:   # the profiler must never sink the measurement
            notes.append(f"pass {'+'.join(counters)}: {ex}")
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for ent in agg.values():
        ent["dispatches"] = max(ent.pop("_n").values())
    return (agg or None), "; ".join(notes)


def pmc_traffic(agg):
    """per phase / kernel family, per bench step: fabric bytes (FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE doubled: see
    FETCH_SIZE_SCALE), the bytes as the counters tally them, L2 and fabric read requests, L2 hits and misses when collected"""
    out = {}
    for ph, names in PHASE_KERNELS.items():
        tot = raw = req_l2 = req_ea = hit = miss = 0.0
        seen = False
        for kname, ent in agg.items():
            if not _kernel_in(kname, names) or "FETCH_SIZE" not in ent or "WRITE_SIZE" not in ent:
                continue
            seen = True
            tot += (ent["FETCH_SIZE"] * FETCH_SIZE_SCALE + ent["WRITE_SIZE"]) * 1024
            raw += (ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024
            req_l2 += ent.get("TCP_TCC_READ_REQ_sum", 0.0)
            req_ea += ent.get("TCC_EA0_RDREQ_sum", 0.0)
            hit += ent.get("TCC_HIT_sum", 0.0)
            miss += ent.get("TCC_MISS_sum", 0.0)
        if seen:
            out[ph] = {"bytes": int(tot), "bytes_raw": int(raw), "l2_read_requests": int(req_l2) or None, "hbm_read_requests": int(req_ea) or None}
            if hit + miss > 0:
                out[ph].update({"l2_hits": int(hit), "l2_misses": int(miss), "l2_hit_rate": round(hit / (hit + miss), 4)})
    return out


def committed_traffic(tag):
    """profiles/traffic_per_launch.json, per kernel only while the sources it was measured on are unchanged"""
    path = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
    try:
        ent = json.load(open(path)).get(tag, {})
    except Exception:
        return {}, {}
    fresh, stale = {}, {}
    for ph, srcs in KERNEL_SOURCES.items():
        if ph not in ent:
            continue
        rec = ent[ph] if isinstance(ent[ph], dict) else {"bytes": ent[ph]}
        (fresh if ent.get("_stamp", {}).get(ph) == source_stamp(srcs) else stale)[ph] = rec
    return fresh, stale
