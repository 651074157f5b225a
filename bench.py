#!/usr/bin/env python3
"""bench.py -- M paired-reads/s through the whole sketch->peak path (phases A->D), index and
packed reads resident in HBM, on N GPUs of one node (one process per GPU, RCCL over xGMI).

A step = counts_clear -> count_kmers (A) -> [count-table exchange] -> ref_scan (B) -> vote (C)
-> [vote all-reduce] -> write_intervals (D) over a synthetic workload of BASELINE.json.  Default =
configs[2], the configuration the metric is quoted on ("UHGG-scale ref") and the largest that fits one
GPU: 13 Gbase reference (13000 x 1 Mbp, 156 GB of index resident in HBM), 100 M 150 bp pairs PER GPU
(weak scaling: read shards are independent), k=32 e=3, every read kept (--sample 1).
`--workload 1g` = configs[1] (1 Gbase, 10 M pairs).  Prints ONE JSON line on rank 0.

`python3 bench.py --gpus N` is self-contained: without WORLD_SIZE in the environment it starts its N ranks itself (child
processes, before this process touches a GPU); under `torch.distributed.run` it is one of the ranks.

What the line holds besides the contract's fields (all of it measured inside this run):
  roofline    the dominant kernel among phase A's family, ref_flags and the vote kernel: launch time by HIP events; bytes from
              rocprofv3 --pmc passes of this same command (child processes, before the timed run).  Three fractions of 8 TB/s:
              frac_raw (FETCH_SIZE + WRITE_SIZE as counted), frac_fabric = frac (FETCH_SIZE x 2: every read request of these
              kernels is a 128-B line fill tallied at 64 B -- a calibrated estimate of fabric bytes, Infinity-Cache hits included)
              and frac_model (SURVEY 8d's algorithmic bytes; `model_exceeded` where the code avoids the probes the model prices);
              infinity_cache_share: upper bound (by capacity) of the "HBM" lines the 256 MiB MALL serves (tools/probe_shapes mall)
  at N > 1    value = the REPLICATED form (every rank scans the whole reference: per-GPU work fixed, which is what "weak" means);
              `sharded_index` = the same step with phase B sharded over the ranks (per-GPU work shrinks: not a scaling figure);
              n1_equivalent_ms = this rank's step without the exchanges; rank 0 measures its kernels' traffic live (N = 1 children)
  compulsory  bytes a step cannot avoid reading (reads twice, index, tables) as a fraction of peak
  secondary   configs[1] (the workload where votes, judge and the dense vote kernel do real work), the UHGG reference under a
              sample of 1000 genomes, and the CLI's default --sample 2e9 mode
  e2e         pairs/s from FASTQ files in the page cache through the drop-in entry point (parse, H2D, pack, A-D)
  cpu_baseline  the CPU restatement on the host cores, bounded sample
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PAIR = lambda L, k, e: 2 * (L - k + 1) * e * 64 + (2 * L + 3) // 4   # one 64 B sector per probe + packed bases
HBM_PEAK_GBS = 8000.0                                                               # MI355X_MICROARCH.md: 8 TB/s spec
METRIC = "M paired-reads/s k-mer sketch->peak, UHGG-scale ref; %HBM roofline @1/2/4/8 GPU"
# request-rate ceilings of the memory system, measured by tools/probe_rates.hip (profiles/r01_probe_rates_microbench.txt,
# profiles/r02_probe_shapes.txt): random 4-byte loads that miss to HBM / that hit in L2
CEIL_HBM_GREQ, CEIL_L2_GREQ = 56.0, 254.0
KERNEL_SOURCES = {   # which sources a kernel's measured traffic depends on (stamp of profiles/traffic_per_launch.json)
    "count_A": ("k_count_part.hip", "k_count.hip"), "ref_flags": ("k_scan.hip",), "vote_kernel": ("k_vote.hip",),
}
COMMON_SOURCES = ("lhgt_hash.hpp", "lhgt_common.hpp", "k_ingest.hip", "k_synth.hip")
# FETCH_SIZE tallies 64 B per fabric read request (TCC_EA0_RDREQ), but on gfx950 EVERY request of these kernels is a 128-B line
# fill: the guide says so for wide streaming reads, and tools/probe_shapes.hip calibrates it for random 4-byte probes (two probes
# in the two 64-B halves of one line cost ONE request: split128, 1.07 requests per pair; a 16 B/lane stream shows 128.0 B per
# request; TCC_BUBBLE and the 32-B request counter are zero) -- profiles/r02/probe_shapes_pmc.txt.  So read bytes = FETCH_SIZE x 2.
FETCH_SIZE_SCALE = 2
PHASE_KERNELS = {
    "count_A": ("part_scatter_reads", "part_scatter_keys", "part_apply", "count_direct"),
    "ref_flags": ("ref_flags=", "ref_flags_lite=", "ref_flags_trio="),      # "=": the whole name (ref_flags_fill belongs to the few unsettled tiles)
    "vote_kernel": ("vote_kernel",),
}


def _kernel_in(kname, names):
    return any(kname == n[:-1] if n.endswith("=") else kname.startswith(n) for n in names)


def source_stamp(names):
    h = hashlib.sha256()
    for n in sorted(set(names) | set(COMMON_SOURCES)):
        with open(os.path.join(ROOT, "localhgt_amd", "csrc", n), "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------- launcher
def self_launch(args, argv):
    """--gpus N without a launcher: start the N ranks as children (nothing here has touched a GPU yet)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---------------------------------------------------------------------------------------------- PMC (child processes)
def collect_pmc(args, passes, timeout_s=420):
    """Run this same command (1 step, no extras) under `rocprofv3 --pmc`, one child per counter group, and return
    {kernel base name: {counter: sum over the step's dispatches, 'dispatches': n}}.  Must run before this process touches
    the GPU.  Counters only (no trace domains besides the kernel dispatch records the CSV needs)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    agg, notes = {}, []
    base = [sys.executable if os.path.basename(sys.executable).startswith("python") else "python3", os.path.abspath(__file__),
            "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--no-pmc", "--no-verify",
            "--workload", args.workload, "--pairs", str(args.pairs), "--contigs", str(args.contigs),
            "--contig-len", str(args.contig_len), "-k", str(args.k), "-e", str(args.e), "--count-mode", str(args.count_mode),
            "--debug", str(args.debug), "--sample-contigs", str(args.sample_contigs), "--ref-form", args.ref_form] + (["--ragged"] if args.ragged else [])
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
    """HBM bytes per bench step per phase/kernel family from a collect_pmc() summary (FETCH_SIZE / WRITE_SIZE count KiB;
    FETCH_SIZE doubled: see FETCH_SIZE_SCALE)"""
    out = {}
    for ph, names in PHASE_KERNELS.items():
        tot, raw, req_l2, req_ea, seen = 0.0, 0.0, 0.0, 0.0, False
        for kname, ent in agg.items():
            if not _kernel_in(kname, names) or "FETCH_SIZE" not in ent or "WRITE_SIZE" not in ent:
                continue
            seen = True
            tot += (ent["FETCH_SIZE"] * FETCH_SIZE_SCALE + ent["WRITE_SIZE"]) * 1024
            raw += (ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024
            req_l2 += ent.get("TCP_TCC_READ_REQ_sum", 0.0)
            req_ea += ent.get("TCC_EA0_RDREQ_sum", 0.0)
        if seen:
            out[ph] = {"bytes": int(tot), "bytes_raw": int(raw), "l2_read_requests": int(req_l2) or None, "hbm_read_requests": int(req_ea) or None}
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


# ---------------------------------------------------------------------------------------------- files for e2e / cpu legs
def write_fasta(path, ref, n_contigs, contig_len):
    with open(path, "wb") as f:
        for c in range(n_contigs):
            f.write(b">g%d\n" % (c + 1))
            f.write(ref[c * contig_len:(c + 1) * contig_len].tobytes())
            f.write(b"\n")


def write_fastq(path, mate, n, L, suffix):
    """4-line records `@r<9 digits>/<suffix>`, vectorised (4 M records in a second or two)"""
    import numpy as np
    a = mate.reshape(n, L)
    ids = np.char.zfill(np.arange(n).astype("U9"), 9)
    head = np.char.add(np.char.add("@r", ids), "/" + suffix)
    hb = np.frombuffer("".join(head.tolist()).encode(), dtype=np.uint8).reshape(n, -1)
    hl = hb.shape[1]
    rec = np.empty((n, hl + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, :hl] = hb
    rec[:, hl] = 10
    rec[:, hl + 1: hl + 1 + L] = a
    rec[:, hl + 1 + L] = 10
    rec[:, hl + 2 + L] = ord("+")
    rec[:, hl + 3 + L] = 10
    rec[:, hl + 4 + L: hl + 4 + 2 * L] = ord("I")
    rec[:, hl + 4 + 2 * L] = 10
    with open(path, "wb") as f:
        f.write(rec.tobytes())


def synth_files(tmp, k, e, n_contigs, contig_len, n_pairs, device, seed_ref=1, seed_reads=2):
    from localhgt_amd.engine import Engine
    with Engine(k, e, device=device) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        ref = eng.synth_reference(seed_ref, n_contigs, contig_len, want_host=True)
        m1, m2 = eng.synth_pairs(seed_ref, seed_reads, n_contigs, contig_len, 0, n_pairs, 150, want_host=True)
    fa, f1, f2 = (os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq"))
    write_fasta(fa, ref, n_contigs, contig_len)
    write_fastq(f1, m1, n_pairs, 150, "1")
    write_fastq(f2, m2, n_pairs, 150, "2")
    return fa, f1, f2


def cpu_baseline(k, e, n_contigs, contig_len, n_pairs, device):
    """Time the CPU oracle (oracle/lhgt_oracle.c, all host cores) on a bounded sample of the same
    synthetic workload.  The oracle is the checker/baseline here, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    from conftest import build_oracle
    orc = oracle_api.Oracle(build_oracle())
    orc.set_pretouch(True)     # table page faults before the phase timers, like the reference's memsets do (E:1416, 1458)
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory(prefix="lhgt_cpu_") as tmp:
        fa, f1, f2 = synth_files(tmp, k, e, n_contigs, contig_len, n_pairs, device)
        cpu_iv, gpu_iv = os.path.join(tmp, "interval.txt"), os.path.join(tmp, "interval.gpu.txt")
        rc, rep = orc.run(f1, f2, fa, cpu_iv, 0.1, 0.08, cores, k, 3000000, e, 1, 1.0)
        if rc != 0:
            return None
        # the same files through the product (the index the CPU run wrote is reused): the baseline is only worth quoting if both
        # sides computed the same thing -- the interval file must be the same, byte for byte
        from localhgt_amd import extract_ref
        rep_g = extract_ref.run(extract_ref.Args(f1, f2, fa, gpu_iv, 0.1, 0.08, 1, k, 3000000, e, 1, 1.0), device=device, log=lambda *x: None)
        identical = open(cpu_iv, "rb").read() == open(gpu_iv, "rb").read() and int(rep.n_peaks) == rep_g["n_peaks"] and \
            int(rep.n_filtered) == rep_g["n_filtered"]
        t = rep.t_count + rep.t_scan + rep.t_vote
        return {"value": round(n_pairs / t / 1e6, 6), "unit": "M paired-reads/s", "cores": cores, "kind": "port",
                "sample": f"{n_pairs} pairs x 150 bp vs {n_contigs} x {contig_len} bp synthetic contigs, k={k} e={e}, "
                          f"phases A+B+C of oracle/lhgt_oracle.c (index build excluded): "
                          f"A {rep.t_count:.2f}s B {rep.t_scan:.2f}s C {rep.t_vote:.2f}s",
                "raw_peaks": int(rep.n_peaks), "filtered_peaks": int(rep.n_filtered), "interval_lines": sum(1 for _ in open(cpu_iv)),
                "identical_to_gpu": bool(identical),
                "gpu_same_files": {"total_s": round(rep_g["total_s"], 3), "raw_peaks": rep_g["n_peaks"], "filtered_peaks": rep_g["n_filtered"],
                                   "kernels_ms": round(rep_g["count_kernel_ms"] + rep_g["scan_kernel_ms"] + rep_g["vote_kernel_ms"], 1)}}


def synth_files_sliced(tmp, k, e, n_contigs, contig_len, n_pairs, device, slice_pairs=4_000_000):
    """like synth_files for inputs of tens of GB: the pairs generated and written slice by slice"""
    from localhgt_amd.engine import Engine
    fa, f1, f2 = (os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq"))
    with Engine(k, e, device=device) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        write_fasta(fa, eng.synth_reference(1, n_contigs, contig_len, want_host=True), n_contigs, contig_len)
        for path in (f1, f2):
            open(path, "wb").close()
        for p0 in range(0, n_pairs, slice_pairs):
            n = min(slice_pairs, n_pairs - p0)
            eng.pairs_clear()
            m1, m2 = eng.synth_pairs(1, 2, n_contigs, contig_len, p0, n, 150, want_host=True)
            for path, m, suf in ((f1, m1, "1"), (f2, m2, "2")):
                part = path + ".part"
                write_fastq(part, m, n, 150, suf)      # read ids restart per slice: the path looks at the first one only
                with open(path, "ab") as dst, open(part, "rb") as src:
                    shutil.copyfileobj(src, dst, 1 << 24)
                os.remove(part)
    return fa, f1, f2


def e2e_from_files(k, e, device, n_contigs=100, contig_len=1_000_000, n_pairs=4_000_000, big_pairs=32_000_000):
    """from FASTQ files in the page cache through the drop-in entry point (localhgt_amd.extract_ref.run: what bin/extract_ref
    calls): line count (+ sampling ratio), index (built in the first run, loaded in the second), parse + H2D + pack with phase A
    behind it, phases B-D, interval file.  -t 10 as `localhgt bkp` passes it: the reference's thread chunks are emulated.
    Two sizes: 4 M pairs (2.5 GB of text: the fixed costs show) and `big_pairs` (20 GB: the reads decide)."""
    from localhgt_amd import extract_ref
    quiet = dict(device=device, log=lambda *x: None)
    with tempfile.TemporaryDirectory(prefix="lhgt_e2e_") as tmp:
        fa, f1, f2 = synth_files(tmp, k, e, n_contigs, contig_len, n_pairs, device)
        a = extract_ref.Args(f1, f2, fa, os.path.join(tmp, "interval.txt"), 0.1, 0.08, 10, k, 300_000_000, e, 1, 1.0)
        reps = [extract_ref.run(a, **quiet) for _ in range(3)]
        built, cached = reps[0], min(reps[1:], key=lambda r: r["total_s"])
        plain = min((extract_ref.run(a, emulate_threads=False, **quiet) for _ in range(2)), key=lambda r: r["total_s"])
        packed = min((extract_ref.run(a, ref_form="packed", **quiet) for _ in range(2)), key=lambda r: r["total_s"])
        fq_bytes = os.path.getsize(f1) + os.path.getsize(f2)
        out = {"value": round(n_pairs / cached["total_s"] / 1e6, 3), "unit": "M paired-reads/s",
               "what": f"extract_ref -t 10 (thread emulation, the CLI default) on {n_pairs} pairs ({fq_bytes / 1e9:.2f} GB of FASTQ, page cache) vs {n_contigs} x {contig_len} bp, "
                       f"k={k} e={e}, cached index; whole call incl. context set-up, index load, parse, H2D, packing, A-D, interval file",
               "total_s": round(cached["total_s"], 3), "ingest_s": round(cached["ingest_s"], 3), "emulated_threads": cached["emulated_threads"],
               "index_load_s": round(cached.get("index_s", 0.0), 3), "reads_s": round(cached.get("reads_s", 0.0), 3),
               "kernels_ms": round(cached["count_kernel_ms"] + cached["scan_kernel_ms"] + cached["vote_kernel_ms"], 1),
               "fastq_GB_per_s": round(fq_bytes / cached["total_s"] / 1e9, 2),
               "without_thread_emulation": {"value": round(n_pairs / plain["total_s"] / 1e6, 3), "total_s": round(plain["total_s"], 3),
                                            "what": "LHGT_EMULATE_THREADS=0: the -t 1 result whatever -t says"},
               "with_index_build": {"value": round(n_pairs / built["total_s"] / 1e6, 3), "total_s": round(built["total_s"], 3)},
               "with_packed_reference": {"value": round(n_pairs / packed["total_s"] / 1e6, 3), "total_s": round(packed["total_s"], 3),
                                         "reference_load_s": round(packed.get("index_s", 0.0), 3), "same_peaks": (packed["n_peaks"], packed["n_filtered"]) == (cached["n_peaks"], cached["n_filtered"]),
                                         "what": "LHGT_REF_FORM=packed: no index file read; the FASTA text goes to the GPU, is stripped and packed there, phase B recomputes the hashes"},
               "raw_peaks": cached["n_peaks"], "filtered_peaks": cached["n_filtered"]}
    if big_pairs and shutil.disk_usage(tempfile.gettempdir()).free > 2.2 * 320 * 2 * big_pairs:
        with tempfile.TemporaryDirectory(prefix="lhgt_e2e_") as tmp:
            t0 = time.time()
            fa, f1, f2 = synth_files_sliced(tmp, k, e, n_contigs, contig_len, big_pairs, device)
            gen_s = time.time() - t0
            fq_bytes = os.path.getsize(f1) + os.path.getsize(f2)
            legs = {}
            for tag, sample, kw in (("sample_1", 1.0, {}), ("sample_1_packed_reference", 1.0, {"ref_form": "packed"}),
                                    ("default_sample_2e9", 2e9, {}), ("default_sample_2e9_packed_reference", 2e9, {"ref_form": "packed"})):
                a = extract_ref.Args(f1, f2, fa, os.path.join(tmp, "interval.txt"), 0.1, 0.08, 10, k, 300_000_000, e, 1, sample)
                r = min((extract_ref.run(a, **dict(quiet, **kw)) for _ in range(3 if not legs else 2)), key=lambda r: r["total_s"])
                legs[tag] = {"value": round(big_pairs / r["total_s"] / 1e6, 2), "unit": "M input pairs/s", "total_s": round(r["total_s"], 3),
                             "reads_s": round(r["reads_s"], 3), "reference_s": round(r["index_s"], 3), "pairs_kept": r["pairs_kept"],
                             "ratio_percent": round(r["ratio"], 4), "raw_peaks": r["n_peaks"], "filtered_peaks": r["n_filtered"],
                             "fastq_GB_per_s": round(fq_bytes / r["total_s"] / 1e9, 1)}
            out["big"] = dict(legs, what=f"the same call on {big_pairs} pairs ({fq_bytes / 1e9:.1f} GB of FASTQ in the page cache, written in {gen_s:.0f} s), -t 10; "
                                         "default_sample_2e9 = the CLI's default --sample 2000000000 (cal_sam_ratio's base count from the line plan, "
                                         "pairs kept by the sampling array)")
    return out


# ---------------------------------------------------------------------------------------------- did the run find what was planted?
_M64 = (1 << 64) - 1


def _mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def planted_breakpoints(n_contigs, contig_len, sample_contigs=0, ref_seed=1, transfer_len=3000):
    """(1-based contig number, position) of every breakpoint the synthetic sample carries (k_synth.hip: transfer_sites): sample
    genome pair i = recipient contig 2i with a 3 kb insert at r0, donor contig 2i+1 that lost [d0, d0 + 3 kb)"""
    n_sample = (sample_contigs & ~1) if 0 < sample_contigs <= n_contigs else (n_contigs // 2) & ~1
    span = contig_len - 3 * transfer_len
    out = []
    for i in range(n_sample // 2):
        h = _mix64((ref_seed * 0x51ED2701 + i) & _M64)
        r0, d0 = transfer_len + h % span, transfer_len + _mix64(h) % span
        out += [(2 * i + 1, r0), (2 * i + 2, d0), (2 * i + 2, d0 + transfer_len)]
    return out


def interval_recall(interval_path, breakpoints):
    """the reference's own quality measure for this stage (paper_results/evaluation.py:64-76): the fraction of true breakpoints
    that fall inside an extracted interval"""
    by_contig = {}
    with open(interval_path) as f:
        for ln in f:
            c, a, b = (int(x) for x in ln.split())
            by_contig.setdefault(c, []).append((a, b))
    hit = sum(1 for c, p in breakpoints if any(a <= p <= b for a, b in by_contig.get(c, ())))
    return {"breakpoints": len(breakpoints), "inside_an_interval": hit, "recall": round(hit / max(1, len(breakpoints)), 4),
            "interval_lines": sum(len(v) for v in by_contig.values())}


# ---------------------------------------------------------------------------------------------- the timed loop
class Workload:
    def __init__(self, eng, dist, rank, world, shard_index, out_path):
        self.eng, self.dist, self.rank, self.world, self.shard_index, self.out_path = eng, dist, rank, world, shard_index, out_path
        self.xch = {"merge_counts": 0.0, "sharded_scan": 0.0, "sum_votes": 0.0}

    def _timed(self, name, fn, *a):
        t0 = time.perf_counter()
        r = fn(*a)
        self.xch[name] += time.perf_counter() - t0
        return r

    def step(self):
        eng, dist = self.eng, self.dist
        eng.counts_clear()
        eng.count_kmers()
        if dist:
            self._timed("merge_counts", dist.merge_counts, eng)
        if self.shard_index:
            n_peaks = self._timed("sharded_scan", dist.sharded_scan, eng, 0.1, 0.08, 300_000_000)
        else:
            n_peaks = eng.ref_scan(0.1, 0.08, 300_000_000)
        eng.vote()
        if dist:
            self._timed("sum_votes", dist.sum_votes, eng)
        nf = eng.write_intervals(self.out_path) if self.rank == 0 else -1
        return n_peaks, nf

    def fence(self):
        import torch
        self.eng.synchronize()
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
            torch.cuda.synchronize()

    def run(self, steps, warmup):
        """W untimed steps, then exactly K timed ones between fences; every step must reproduce the same peaks"""
        import torch
        for _ in range(warmup):
            self.step()
        self.fence()
        for key in self.xch:
            self.xch[key] = 0.0
        t0 = time.time()
        ms = [0.0, 0.0, 0.0, 0.0]
        seen = set()
        for _ in range(steps):
            seen.add(self.step())
            for ph in range(4):
                ms[ph] += self.eng.phase_ms(ph)
        self.fence()
        dt = time.time() - t0
        if self.dist:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dist._dev())
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        if len(seen) != 1:
            raise SystemExit(f"bench: steps disagree on (raw peaks, filtered peaks): {sorted(seen)}")
        n_peaks, nf = seen.pop()
        return dt, [m / steps for m in ms], n_peaks, nf


def verify_forms(eng):
    """one untimed check that the shortcuts of the timed path change nothing: the form of phase B the engine picks (lite on a
    nearly full table) against the exact form, and the vote kernel it picks against the generic kernel without any prefilter --
    whole tables compared through device-side checksums"""
    res = {}
    for name, dbg in (("picked", 0), ("exact", 8192 | 4)):
        eng.set_debug(dbg)
        n = eng.ref_scan(0.1, 0.08, 300_000_000)
        eng.vote()
        res[name] = (n, eng.digest(eng.DIGEST_LOCI), eng.digest(eng.DIGEST_PEAK_KMER), eng.digest(eng.DIGEST_FLAGS, 0b1111100),
                     eng.digest(eng.DIGEST_VOTES))
    eng.set_debug(0)
    if res["picked"] != res["exact"]:
        raise SystemExit(f"bench --verify: the timed forms disagree with the exact ones: {res}")
    return {"ok": True, "raw_peaks": res["exact"][0], "votes_nonzero": res["exact"][4][1],
            "compared": "peak loci, peak_kmer[2^k], flags of every reference position, votes: picked forms vs exact scan + unfiltered generic vote"}


def mall_share():
    """tools/probe_shapes mall (built by __graft_entry__.build): random-probe rates on 128 MiB / 1 GiB / 16 GiB tables (a table inside the
    256 MiB Infinity Cache is probed no faster than a 1 GiB one) and the cache's share of a table by capacity.  None when the binary is not there."""
    exe = os.path.join(ROOT, "tools", "probe_shapes")
    if not os.path.exists(exe):
        return None
    try:
        res = subprocess.run([exe, "mall"], capture_output=True, text=True, timeout=120)
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception:
        return None


def roofline_entry(desc, ms_step, launches, algo_bytes, traffic_rec, source, ceiling, mall=None, table=None):
    """one kernel against the 8 TB/s HBM peak, three ways (none of them is a guarantee of <= 1):
      frac_raw     FETCH_SIZE + WRITE_SIZE as the counters tally them (64 B per read request)
      frac_fabric  FETCH_SIZE x 2 + WRITE_SIZE: every read request of these kernels is a 128-B line fill (calibrated with
                   tools/probe_shapes.hip, profiles/r02/probe_shapes_pmc.txt) -- an ESTIMATE of the bytes that cross the fabric;
                   it counts Infinity-Cache hits as memory reads (TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ on gfx950), see
                   infinity_cache_share.  `frac` and `achieved` are this figure
      frac_model   SURVEY 8d's algorithmic bytes (one 64-B sector per probe) / time / peak; above 1 (`model_exceeded`) where the
                   partition, the L2-resident bitmap or the lite scan avoid the probes the model prices
    plus the request rate against its microbenchmarked ceiling."""
    ent = {"kernel": desc, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(ms_step, 3),
           "launches_per_step": launches, "launch_ms": round(ms_step / launches, 3) if launches else None,
           "algorithmic_bytes_per_step": algo_bytes,
           "sector_model_GBps": round(algo_bytes / (ms_step * 1e-3) / 1e9, 1) if ms_step > 0 else None}
    if ms_step > 0:
        fm = algo_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        ent.update({"frac_model": round(fm, 4), "model_exceeded": bool(fm > 1.0)})
    if traffic_rec and ms_step > 0:
        b = traffic_rec["bytes"]
        ach = b / (ms_step * 1e-3) / 1e9
        ent.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 4), "frac_fabric": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": b // max(1, launches or 1), "traffic_per_step": b, "scale": FETCH_SIZE_SCALE,
                    "traffic_source": source, "model_speedup": round(algo_bytes / b, 2) if b else None})
        if traffic_rec.get("bytes_raw"):
            raw = traffic_rec["bytes_raw"]
            ent.update({"traffic_raw": raw // max(1, launches or 1), "frac_raw": round(raw / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        req = traffic_rec.get(ceiling[0])
        if req:
            g = req / (ms_step * 1e-3) / 1e9
            ent["request_rate"] = {"value": round(g, 1), "unit": "G requests/s", "counter": ceiling[1], "ceiling": ceiling[2],
                                   "frac_of_ceiling": round(g / ceiling[2], 3), "ceiling_source": ceiling[3]}
        if mall and table in ("1GiB", "16GiB"):
            sh = mall.get(f"share_by_capacity_{table}_table")
            if sh is not None:
                ent["infinity_cache_share"] = {"upper_bound": sh, "of": f"line fills of the kernel's {table} probe table: 256 MiB of Infinity Cache / table size. Rates cannot tell "
                                               f"more: random probes into a 128 MiB table (inside the cache) run at {mall.get('gprobes_per_s', {}).get('128MiB')} G/s, into a 1 GiB one at "
                                               f"{mall.get('gprobes_per_s', {}).get('1GiB')} G/s (tools/probe_shapes mall) -- the line rate is the fabric's, a cache hit is no faster",
                                               "frac_hbm_lower_bound": round(ach / HBM_PEAK_GBS * (1.0 - sh), 4)}
    else:
        ent.update({"achieved": None, "frac": None, "traffic": None, "stale": True,
                    "traffic_source": "none: the rocprofv3 --pmc passes failed and profiles/traffic_per_launch.json was measured on other sources"})
    return ent


PMC_PASSES = [["FETCH_SIZE", "TCP_TCC_READ_REQ_sum"], ["WRITE_SIZE", "TCC_EA0_RDREQ_sum"]]
HBM_CEILING = ("hbm_read_requests", "TCC_EA0_RDREQ_sum", CEIL_HBM_GREQ, "tools/probe_shapes.hip: random 4-byte loads from a 1 GiB table, 128-B line fills per second (profiles/r02/probe_shapes_microbench.txt)")
L2_CEILING = ("l2_read_requests", "TCP_TCC_READ_REQ_sum", CEIL_L2_GREQ, "tools/probe_rates.hip: random 4-byte loads from an L2-resident table (profiles/r01_probe_rates_microbench.txt)")


def rooflines(k, e, L, pairs, ref_bases, packed, per_ms, scan, n_peaks, traffic, src, mall):
    """roofline entries of the three kernels (phase A's family as one) of one workload, and which one dominates the step"""
    algo = ALGO_BYTES_PER_PAIR(L, k, e) * pairs           # algorithmic bytes of one scan over this GPU's pairs
    ref_bytes = ref_bases * (64 * e) + (ref_bases // 4 if packed else ref_bases * 4 * e)   # 204 B per base / 0.25 + 192
    n_batches = -(-pairs // (16 << 20))
    n_chunks = -(-pairs // (4 << 20))
    kern = {"count_A": per_ms[0], "ref_flags": per_ms[3], "vote_kernel": per_ms[2]}
    sparse_vote = scan["tiles"] > 0 and n_peaks > 0 and per_ms[2] > 0 and traffic.get("vote_kernel", {}).get("bytes", algo) < algo / 4
    desc = {
        "count_A": (f"phase A kernel family (part_scatter_reads + part_scatter_keys16 + part_apply per <= 4 Mi-pair chunk, {n_chunks} chunks per step; "
                    "count_direct below k = 26): 714 table updates per pair", algo, 3 * n_chunks, HBM_CEILING, "1GiB"),
        "ref_flags": (("ref_flags_lite (phase B on a nearly saturated table: one probe per base until a hash reads 3, all e at every 8th base)"
                       if scan["lite"] else "ref_flags_trio (phase B on a sparse table: probes per base until a hash does not read 3)" if scan["form"] == "trio-first"
                       else "ref_flags (phase B: e random 2-bit table probes + e index words per reference base)")
                      + ", 1 launch per step", ref_bytes, 1, HBM_CEILING, "1GiB"),
        "vote_kernel": (f"vote_kernel (phase C read re-scan: 714 probes per pair, "
                        f"{'answered by the L2-resident bitmap except for its survivors' if sparse_vote else 'into peak_kmer'}), "
                        f"{n_batches} launches per step", algo, n_batches, L2_CEILING if sparse_vote else HBM_CEILING, None if sparse_vote else "16GiB"),
    }
    roof = {ph: roofline_entry(desc[ph][0], kern[ph], desc[ph][2], desc[ph][1], traffic.get(ph) if src else None, src, desc[ph][3], mall, desc[ph][4])
            for ph in kern}
    if k < 32:
        for ent in roof.values():
            ent.pop("infinity_cache_share", None)          # the table sizes above are those of k = 32
    dominant = max(kern, key=kern.get)                      # over A (as one entry), B's probe kernel and C
    return roof, dominant


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["uhgg", "1g"], default="uhgg",
                    help="uhgg = BASELINE configs[2] (13000 x 1 Mbp, 100 M pairs/GPU); 1g = configs[1] (1000 x 1 Mbp, 10 M pairs/GPU)")
    ap.add_argument("--pairs", type=int, default=None, help="read pairs per GPU (overrides the workload's)")
    ap.add_argument("--contigs", type=int, default=None, help="contigs of the synthetic reference (overrides the workload's)")
    ap.add_argument("--contig-len", type=int, default=1_000_000)
    ap.add_argument("--sample-contigs", type=int, default=0, help="contigs the synthetic sample is drawn from (0 = half of the reference)")
    ap.add_argument("--ragged", action="store_true", help="the same base stream cut into ~118 k contigs of a catalogue-like length distribution (localhgt_amd.synth.ragged_cuts)")
    ap.add_argument("-k", type=int, default=32)
    ap.add_argument("-e", type=int, default=3)
    ap.add_argument("--shard-index", action="store_true", help="N > 1: time ONLY the reference-sharded phase B (each rank holds 1/N of the index)")
    ap.add_argument("--replicate-index", action="store_true", help="N > 1: time ONLY the replicated form (the whole index on every GPU, no exchange in phase B)")
    ap.add_argument("--ref-form", choices=["index", "packed"], default="index",
                    help="resident form of the reference: the index file's hashes (12 B/base at e=3; what configs[2] names) or the packed bases "
                         "(3/8 B/base), phase B recomputing the hashes")
    ap.add_argument("--count-mode", type=int, default=-1, help="-1 = engine default (adaptive), 0 = direct CAS kernel, 1 = radix partition")
    ap.add_argument("--debug", type=int, default=0, help="engine debug/A-B switches (include/localhgt_hip.h: lhgt_set_debug)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL exchange code even at world size 1 (self-test of the N>1 path)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: the exchanges staged through host memory (localhgt_amd/dist.py) -- for ranks that share a GPU (rehearsal of the N > 1 path on a one-GPU box) and for --dry-run")
    ap.add_argument("--dry-run", action="store_true", help="launcher + process group + exchanges on host tensors, no GPU and no measurement (CPU self-test of the N>1 plumbing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=400_000)
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads and the from-FASTQ leg")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect HBM traffic with rocprofv3 --pmc child runs")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed picked-forms-vs-exact-forms check")
    ap.add_argument("--pmc-out", default=None, help="also write the PMC summary of this run to this JSON file")
    args = ap.parse_args()
    wl_contigs, wl_pairs = (13000, 100_000_000) if args.workload == "uhgg" else (1000, 10_000_000)
    args.contigs = args.contigs or wl_contigs
    args.pairs = args.pairs or wl_pairs
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world, local)

    k, e, L = args.k, args.e, 150
    workload_tag = (f"{args.contigs}x{args.contig_len}_{args.pairs}_k{k}_e{e}" + (f"_s{args.sample_contigs}" if args.sample_contigs else "")
                    + ("_packed" if args.ref_form == "packed" else "") + ("_ragged" if args.ragged else ""))
    # ---- measured HBM traffic: child runs of this command under rocprofv3 --pmc, before this process touches the GPU.  At N > 1
    # rank 0 does the same with N = 1 children (every rank runs phases A and C on a read shard of the N = 1 size and shape, and the
    # replicated phase B on the whole reference: its kernels are the N = 1 kernels) while the other ranks wait at the rendezvous
    pmc, pmc_note, traffic, traffic_src = None, "", {}, None
    mall = None
    if rank == 0 and not args.no_pmc and not args.force_dist:
        t0 = time.time()
        pmc, pmc_note = collect_pmc(args, PMC_PASSES)
        if pmc:
            traffic = pmc_traffic(pmc)
            traffic_src = (f"rocprofv3 --pmc passes of this run ({time.time() - t0:.0f} s, 1 step each; FETCH_SIZE+TCP_TCC_READ_REQ, WRITE_SIZE+TCC_EA0_RDREQ)"
                           + (", as N = 1 children of rank 0 on this rank's per-GPU workload" if world > 1 else ""))
            if args.pmc_out:
                json.dump({"tag": workload_tag, "kernels": pmc, "per_step": traffic,
                           "_stamp": {ph: source_stamp(s) for ph, s in KERNEL_SOURCES.items()}}, open(args.pmc_out, "w"), indent=1, sort_keys=True)
        mall = mall_share()
    extra_traffic = {}
    extras = world == 1 and not args.no_extras and not args.force_dist and not args.debug
    if extras and not args.no_pmc and (args.contigs, args.pairs, args.sample_contigs, args.ragged) == (13000, 100_000_000, 0, False):
        for tag, over in (("1g", dict(workload="1g", contigs=1000, pairs=10_000_000)), ("deep", dict(sample_contigs=300)), ("ragged", dict(ragged=True))):
            pm, note = collect_pmc(argparse.Namespace(**dict(vars(args), **over)), PMC_PASSES)
            if pm:
                extra_traffic[tag] = pmc_traffic(pm)
            pmc_note = "; ".join(x for x in (pmc_note, note) if x)
    if rank == 0:
        fresh, stale = committed_traffic(workload_tag)
        for ph, rec in fresh.items():
            if ph not in traffic:
                traffic[ph] = rec
                traffic_src = traffic_src or "profiles/traffic_per_launch.json (measured on these sources; the live rocprofv3 passes failed)"

    import torch
    from localhgt_amd.engine import Engine
    local = local % max(1, torch.cuda.device_count())          # --backend gloo: ranks may share a GPU
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        from localhgt_amd.dist import Exchange
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(key, val)
        dist = Exchange.from_env(backend=args.backend)

    eng = Engine(k, e, device=local)
    eng.rng_seed(1)
    eng.coder_generate()
    if args.count_mode >= 0:
        eng.set_count_mode(args.count_mode)
    if args.debug:
        eng.set_debug(args.debug)
    if args.ref_form == "packed":
        eng.set_reference_form(True)
    out_path = os.path.join(tempfile.gettempdir(), f"lhgt_bench_interval_{os.getpid()}.txt")

    def load_reference(sharded):
        if args.ragged:
            from localhgt_amd.synth import ragged_cuts
            if sharded:
                raise SystemExit("bench: --ragged is an N = 1 / replicated workload")
            eng.synth_reference_cuts(1, args.contigs, args.contig_len, ragged_cuts(args.contigs * args.contig_len))
        elif sharded:
            eng.synth_reference_shard(1, args.contigs, args.contig_len, rank, world)  # this rank's contig range only
        else:
            eng.synth_reference(1, args.contigs, args.contig_len)                   # whole reference resident in HBM

    # N > 1: SURVEY.md 8e lays phase B out sharded by contig range (each rank scans 1/N of the reference, the peaks are exchanged).
    # That makes the per-GPU work SHRINK with N, so it cannot be the "weak scaling" figure: the timed K steps -- `value` -- run the
    # replicated form (per-GPU work fixed: its own reads, the whole reference); the sharded form is measured next to it.
    forms = [False] if dist is None or world == 1 and not args.shard_index else [False, True]
    if args.replicate_index:
        forms = [False]
    if args.shard_index and dist is not None:
        forms = [True]
    t0 = time.time()
    load_reference(forms[0])
    eng.synth_options(0, 20, args.sample_contigs)
    eng.synth_pairs(1, 2, args.contigs, args.contig_len, rank * args.pairs, args.pairs, L)   # this rank's shard, packed, resident
    eng.synchronize()
    setup_s = time.time() - t0
    wl = Workload(eng, dist, rank, world, forms[0], out_path)

    verify = None
    if world == 1 and not args.no_verify and not args.debug:
        eng.counts_clear()
        eng.count_kmers()
        verify = verify_forms(eng)
    dt, per_ms, n_peaks, nf = wl.run(args.steps, args.warmup)
    scan = eng.scan_info()
    xch_main = dict(wl.xch)
    other_form = None
    if len(forms) > 1:                                   # the other form of phase B, a few steps, same reads
        load_reference(forms[1])
        wl2 = Workload(eng, dist, rank, world, forms[1], out_path)
        st2 = min(args.steps, 5)
        dt2, per2, n2, nf2 = wl2.run(st2, 1)
        other_form = {"form": "reference-sharded phase B" if forms[1] else "replicated phase B", "value": round(args.pairs * world * st2 / dt2 / 1e6, 4),
                      "unit": "M paired-reads/s", "ms_per_step": round(dt2 / st2 * 1e3, 3), "steps": st2,
                      "phase_ms": {"count_A": round(per2[0], 3), "scan_B": round(per2[1], 3), "vote_C": round(per2[2], 3)},
                      "exchange_ms": {kk: round(v / st2 * 1e3, 3) for kk, v in wl2.xch.items()},
                      "same_peaks": (n2, nf2) == (n_peaks, nf),
                      "note": "each rank scans 1/N of the reference: per-GPU work shrinks with N, so this is NOT a weak-scaling figure -- "
                              "value/ms_per_step of the line are the replicated form's"}
    if rank != 0:
        eng.close()
        if dist:
            dist.close()
        return
    total_pairs = args.pairs * world * args.steps
    per = {"count_A": per_ms[0], "scan_B": per_ms[1], "vote_C": per_ms[2]}   # HIP events on the engine stream
    ref_bases = args.contigs * args.contig_len
    if forms[0]:
        ref_bases = (args.contigs * (rank + 1) // world - args.contigs * rank // world) * args.contig_len   # this rank's contig range
        if world > 1 and traffic.get("ref_flags"):
            share = ref_bases / (args.contigs * args.contig_len)
            traffic["ref_flags"] = {kk: (int(v * share) if isinstance(v, (int, float)) else v) for kk, v in traffic["ref_flags"].items()}
    roof, dominant = rooflines(k, e, L, args.pairs, ref_bases, args.ref_form == "packed", per_ms, scan, n_peaks, traffic, traffic_src, mall)
    # bytes a step cannot avoid: the packed reads twice (A and C), the resident index once, count table written and read,
    # peak_kmer cleared (E:1458) -- everything else is the price of random access
    read_store = args.pairs * 2 * 3 * ((L + 31) // 32 + 1) * 4
    compulsory = 2 * read_store + (ref_bases * 3 // 8 if args.ref_form == "packed" else ref_bases * 4 * e) + 2 * ((1 << k) // 4) + (1 << k) * 4
    step_s = dt / args.steps
    xch_ms = {kk: round(v / args.steps * 1e3, 3) for kk, v in xch_main.items()} if dist else None
    cfg_no = 2 if (args.contigs, args.pairs, args.sample_contigs, args.ragged) == (13000, 100_000_000, 0, False) else \
        1 if (args.contigs, args.pairs, args.sample_contigs, args.ragged) == (1000, 10_000_000, 0, False) else "-"
    line = {
        "metric": METRIC, "value": round(total_pairs / dt / 1e6, 4), "unit": "M paired-reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{cfg_no}]: "
                               f"{args.contigs}x{args.contig_len} bp synthetic ref ({args.contigs * args.contig_len / 1e9:.2f} Gbase{', cut into a ragged catalogue' if args.ragged else ''}, "
                               f"{'packed bases resident, hashes recomputed' if args.ref_form == 'packed' else 'index resident'}), {args.pairs} 150bp pairs per GPU, k={k} e={e}, sample=1, phases A-D",
                   "pairs_per_gpu": args.pairs, "ref_bases": args.contigs * args.contig_len, "k": k, "e": e,
                   "parallelism": f"reads sharded x{world}" + (", index sharded" if forms[0] else ", phase B replicated on every GPU (per-GPU work fixed)" if world > 1 else "")},
        "world_size": torch.distributed.get_world_size() if dist else 1, "backend": dist.backend if dist else None,
        "phase_ms": {kk: round(v, 3) for kk, v in per.items()},
        "exchange_ms": xch_ms,
        "n1_equivalent_ms": round(step_s * 1e3 - sum(xch_ms.values()), 3) if xch_ms else round(step_s * 1e3, 3),
        "sharded_index" if (other_form and forms[1]) else "replicated_index": other_form,
        "scan_B_form": scan,
        "raw_peaks": n_peaks, "filtered_peaks": nf, "setup_s": round(setup_s, 2),
        "planted_transfers": interval_recall(out_path, planted_breakpoints(args.contigs, args.contig_len, args.sample_contigs)) if world == 1 and not args.ragged else None,
        "verify": verify,
        "roofline": dict(roof[dominant], kernel=roof[dominant]["kernel"] + " -- the dominant kernel of this workload"),
        "roofline_other": {ph: roof[ph] for ph in roof if ph != dominant},
        "infinity_cache": mall,
        "compulsory": {"bytes_per_step": compulsory, "frac_of_peak": round(compulsory / step_s / (HBM_PEAK_GBS * 1e9), 4),
                       "what": "packed reads twice + resident index once + count table written and read + peak_kmer cleared; a step at HBM peak would take "
                               f"{compulsory / (HBM_PEAK_GBS * 1e9) * 1e3:.0f} ms"},
        "note": "roofline.frac = frac_fabric = (2 x FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s: every fabric read request on gfx950 is a 128-B line fill "
                "tallied at 64 B (calibrated by tools/probe_shapes.hip, profiles/r02/), so this is an estimate of fabric bytes, Infinity-Cache hits included "
                "(infinity_cache_share); frac_raw uses the counters as they are; frac_model is SURVEY 8d's one-64-B-sector-per-probe figure and exceeds 1 "
                "(model_exceeded) where partitioning, the L2 bitmap or the lite scan avoid the probes. request_rate: line fills (or L2 requests) per second "
                "against the microbenchmarked ceiling (56 G/s HBM lines, 254 G/s L2); DESIGN.md 4-5"
                + (f"; pmc: {pmc_note}" if pmc_note else ""),
    }
    eng.pairs_clear()
    if extras:
        args.headline_peaks = (n_peaks, nf)
        line["secondary"] = secondary_workloads(eng, args, wl, local, extra_traffic, mall)
    eng.close()
    if dist:
        dist.close()
    if extras:
        try:
            line["e2e"] = e2e_from_files(k, e, local)
        except Exception as ex:
            line["e2e"] = {"error": str(ex)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(k, e, 20, 1_000_000, args.cpu_pairs, local)
        except Exception as ex:  # the baseline must never sink the measurement
            line["cpu_baseline"] = {"error": str(ex)}
    # RCCL writes its version banner through C stdio, which on a pipe is flushed at exit, i.e. after Python's output:
    # flush it now so that the JSON line is the last thing on stdout
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(line), flush=True)
    try:
        os.remove(out_path)
    except OSError:
        pass
    if isinstance(line.get("cpu_baseline"), dict) and line["cpu_baseline"].get("identical_to_gpu") is False:
        sys.exit("bench: the GPU path and the CPU baseline wrote different interval files for the same inputs")


def secondary_workloads(eng, args, wl, local, extra_traffic, mall):
    """the other regimes of the same path, a few steps each (N = 1): results a reader needs next to the headline, whose
    synthetic sample (half of a 13 Gbase reference) saturates the 2^32-slot table and yields no voted peak"""
    from localhgt_amd.engine import Engine
    out = {}
    k, e, L = args.k, args.e, 150
    live = "rocprofv3 --pmc passes of this run on this workload"

    def leg(engine, pairs, steps=3, n_contigs=None, sample_contigs=0, traffic=None, ref_bases=None, packed=False, recall=True):
        w = Workload(engine, None, 0, 1, False, wl.out_path)
        dt, per_ms, n_peaks, nf = w.run(steps, 1)
        d = {"value": round(pairs * steps / dt / 1e6, 3), "unit": "M paired-reads/s", "ms_per_step": round(dt / steps * 1e3, 2),
             "phase_ms": {"count_A": round(per_ms[0], 2), "scan_B": round(per_ms[1], 2), "vote_C": round(per_ms[2], 2)},
             "scan_B_form": engine.scan_info(), "raw_peaks": n_peaks, "filtered_peaks": nf, "steps": steps, "pairs": pairs}
        if recall:
            d["planted_transfers"] = interval_recall(wl.out_path, planted_breakpoints(n_contigs or args.contigs, args.contig_len, sample_contigs))
        if traffic:
            roof, dom = rooflines(k, e, L, pairs, ref_bases or (n_contigs or args.contigs) * args.contig_len, packed, per_ms, d["scan_B_form"], n_peaks,
                                  traffic, live, mall)
            d["roofline"] = dict(roof[dom], kernel=roof[dom]["kernel"] + " -- the dominant kernel of this workload")
            d["roofline_other"] = {ph: {kk: v for kk, v in r.items() if kk in ("ms_per_step", "frac", "frac_raw", "frac_model", "model_exceeded", "request_rate", "infinity_cache_share")}
                                   for ph, r in roof.items() if ph != dom}
        return d

    headline = (args.contigs, args.pairs, args.sample_contigs, args.ragged) == (13000, 100_000_000, 0, False)
    try:
        if headline:
            # the same reference under a sample of 300 of its genomes at 10x: 0.3 G distinct k-mers x 3 hashes leave the 2^32 slots
            # four fifths empty, transfers are found and voted -- what the algorithm is built for.  (From 1000 genomes up the
            # sample's k-mers alone saturate half of the slots, every window of the whole reference turns "good" and 2.4e8
            # noise peaks appear -- 3 to 7 s per step -- which is what the reference's own down-sampling exists to avoid.)
            fp = 10_000_000
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, args.contigs, args.contig_len, 0, fp, L)
            out["uhgg_focused_sample"] = dict(leg(eng, fp, sample_contigs=300), workload="13000x1000000 bp ref, 10 M pairs drawn from 300 of its contigs (a metagenome holds few of a catalogue's genomes; 10x), sample=1")
            eng.pairs_clear()
            # ... and DEEP: the headline's 100 M pairs from those 300 genomes (100x): the table a fifth full, trio-first scan,
            # transfers found, the dense-ish vote -- the realistic counterpart of the headline's read count
            eng.synth_pairs(1, 2, args.contigs, args.contig_len, 0, args.pairs, L)
            out["uhgg_deep_focused_sample"] = dict(leg(eng, args.pairs, sample_contigs=300, traffic=extra_traffic.get("deep")),
                                                   workload="13000x1000000 bp ref, 100 M pairs drawn from 300 of its contigs (100x), sample=1")
            eng.pairs_clear()
            # the CLI's default --sample 2000000000 (E:1392-1398): 2e9 / (2 * 100 M * 150) = 6.67 % of the pairs survive the
            # sampling array; any subset of iid pairs is iid, so the kept pairs are generated directly
            kept = int(2e9 / (2 * 150))
            eng.synth_options(0, 20, 0)
            eng.synth_pairs(1, 2, args.contigs, args.contig_len, 0, kept, L)
            d = leg(eng, kept)
            d.update(workload=f"configs[2] under the pipeline's default --sample 2000000000: {kept} of 100 M pairs kept (resident; a real run is bound by parsing the other 93 %)",
                     input_pairs=args.pairs, input_pairs_per_s_M=round(args.pairs / (d["ms_per_step"] * 1e-3) / 1e6, 1))
            out["uhgg_default_sample"] = d
            eng.pairs_clear()
            # a RAGGED catalogue: the same 13 Gbase cut into ~118 k contigs (median 4.8 kb, a third shorter than one scan tile) -- what
            # UHGG looks like.  The k - 1 positions without a k-mer at every contig end are contrast peaks (E:931-932, 644-671), so
            # ten times as many peaks register k-mers and phase C is the phase that feels it
            from localhgt_amd.synth import ragged_cuts
            cuts = ragged_cuts(args.contigs * args.contig_len)
            eng.synth_reference_cuts(1, args.contigs, args.contig_len, cuts)
            eng.synth_pairs(1, 2, args.contigs, args.contig_len, 0, args.pairs, L)
            d = leg(eng, args.pairs, traffic=extra_traffic.get("ragged"), recall=False)
            d.update(workload=f"configs[2]'s bases and reads, the reference cut into {len(cuts) - 1} pieces of a catalogue-like length distribution (localhgt_amd.synth.ragged_cuts)")
            out["uhgg_ragged_reference"] = d
            eng.pairs_clear()
            # ... and the ragged catalogue under the deep focused sample (100 M pairs from 300 Mbase of it, about 2700 of its contigs):
            # what a real run on a real catalogue looks like -- contig ends are peaks only where the sample covers them
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, args.contigs, args.contig_len, 0, args.pairs, L)
            d = leg(eng, args.pairs, recall=False)
            d.update(workload="the ragged catalogue under the deep focused sample: 100 M pairs drawn from the first 300 Mbase of its base stream (100x), sample=1")
            out["uhgg_ragged_deep_focused"] = d
            eng.synth_options(0, 20, 0)
            eng.pairs_clear()
            eng.synth_reference(1, args.contigs, args.contig_len)
            if args.ref_form == "index":
                # the headline workload once more with the reference resident as packed bases (3/8 byte per base instead of the
                # index file's 12): phase B recomputes the hashes instead of streaming them -- same peaks, 32x less resident
                index_bytes = eng.reference_info()["resident_bytes"]
                eng.set_reference_form(True)
                eng.synth_reference(1, args.contigs, args.contig_len)
                eng.synth_pairs(1, 2, args.contigs, args.contig_len, 0, args.pairs, L)
                d = leg(eng, args.pairs)
                d.update(workload="configs[2] with the reference resident as packed bases (lhgt_set_reference_form(1), LHGT_REF_FORM=packed), hashes recomputed in phase B",
                         resident_reference_bytes=eng.reference_info()["resident_bytes"], resident_index_bytes=index_bytes,
                         same_peaks_as_headline=(d["raw_peaks"], d["filtered_peaks"]) == tuple(args.headline_peaks))
                out["uhgg_packed_reference"] = d
                eng.pairs_clear()
    except Exception as ex:
        out["uhgg_error"] = str(ex)
    eng.close()
    try:
        if headline:
            out["pipelined_samples"] = pipelined_samples(k, e, local, args.contigs, args.contig_len, args.pairs)
    except Exception as ex:
        out["pipelined_error"] = str(ex)
    try:
        if (args.contigs, args.pairs) != (1000, 10_000_000):
            with Engine(k, e, device=local) as e1:
                e1.rng_seed(1)
                e1.coder_generate()
                e1.synth_reference(1, 1000, 1_000_000)
                e1.synth_pairs(1, 2, 1000, 1_000_000, 0, 10_000_000, L)
                d = leg(e1, 10_000_000, steps=5, n_contigs=1000, traffic=extra_traffic.get("1g"))
                d["workload"] = "BASELINE configs[1]: 1000x1000000 bp ref, 10 M pairs, k=32 e=3, sample=1"
                out["configs1_1g"] = d
    except Exception as ex:
        out["configs1_error"] = str(ex)
    try:
        if headline:
            # BASELINE configs[4] names a reference of more than 50 GB, 200 M reads over 8 GPUs, k = 21 / 32.  Its index (12 bytes per
            # base: 600 GB) only fits sharded over the node; packed (3/8 byte per base) the whole 50 Gbase reference, its per-position
            # arrays and the tables fit ONE GPU.  One GPU's share of the reads (25 M pairs) drawn from 300 of the 50 000 genomes.
            nc, fp = 50_000, 25_000_000
            legs = {}
            for kk in (32, 21):
                with Engine(kk, e, device=local) as e5:
                    e5.rng_seed(1)
                    e5.coder_generate()
                    e5.set_reference_form(True)
                    e5.synth_reference(1, nc, args.contig_len)
                    e5.synth_options(0, 20, 300)
                    e5.synth_pairs(1, 2, nc, args.contig_len, 0, fp, L)
                    d = leg(e5, fp, steps=2, n_contigs=nc, sample_contigs=300)
                    d["resident_reference_bytes"] = e5.reference_info()["resident_bytes"]
                    legs[f"k{kk}"] = d
            out["configs4_progenomes_1gpu"] = dict(legs, workload=f"{nc}x{args.contig_len} bp ref (50 Gbase) resident as packed bases on ONE GPU, 25 M pairs "
                                                                "(one GPU's share of configs[4]'s 200 M) from 300 of its genomes, e=3, sample=1, k = 32 and 21")
    except Exception as ex:
        out["configs4_error"] = str(ex)
    return out


def pipelined_samples(k, e, device, n_contigs, contig_len, pairs, n_samples=4):
    """Many samples against one resident reference is how this path is used, and its phases sit on three different ceilings --
    A on LDS atomics and instruction issue, B's probe kernel on HBM lines, the sparse vote on L2 requests.  Two contexts on one GPU
    (each with its own stream, tables and read store; the reference resident as packed bases in both), two host threads: a sample's
    phase A may run while the other context is in its phases B-D, never two of the same kind at once.  Reported: pairs/s over
    n_samples samples against the same samples one after the other, per-phase kernel times in both modes, and whether every sample's
    peaks and vote table are the serial run's."""
    import threading
    from localhgt_amd.engine import Engine
    engs = []
    for i in range(2):
        g = Engine(k, e, device=device)
        g.rng_seed(1)
        g.coder_generate()
        g.set_reference_form(True)
        g.synth_reference(1, n_contigs, contig_len)
        g.synth_pairs(1, 2 + i, n_contigs, contig_len, 0, pairs, 150)     # two different samples of the same shape
        engs.append(g)

    def sample(g, lock_a, lock_b, rec):
        with lock_a:
            g.counts_clear()
            g.count_kmers()
            a = g.phase_ms(0)
        with lock_b:
            n = g.ref_scan(0.1, 0.08, 300_000_000)
            g.vote()
            rec.append((n, g.digest(g.DIGEST_VOTES), g.digest(g.DIGEST_PEAK_KMER), a, g.phase_ms(1), g.phase_ms(2)))

    class _NoLock:
        def __enter__(self): return self
        def __exit__(self, *a): return False

    for g in engs:                                        # warm-up: allocations, first-touch
        sample(g, _NoLock(), _NoLock(), [])
    serial = [[], []]
    t0 = time.time()
    for s_i in range(n_samples):
        sample(engs[s_i % 2], _NoLock(), _NoLock(), serial[s_i % 2])
    for g in engs:
        g.synchronize()
    t_serial = time.time() - t0
    piped = [[], []]
    la, lb = threading.Lock(), threading.Lock()

    def worker(i):
        for _ in range(n_samples // 2):
            sample(engs[i], la, lb, piped[i])

    t0 = time.time()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in engs:
        g.synchronize()
    t_piped = time.time() - t0
    same = all([r[:3] for r in serial[i]] == [r[:3] for r in piped[i]] for i in range(2))
    for g in engs:
        g.close()

    def mean(recs, j):
        v = [r[j] for rr in recs for r in rr]
        return round(sum(v) / max(1, len(v)), 1)

    return {"samples": n_samples, "pairs_per_sample": pairs,
            "serial": {"value": round(n_samples * pairs / t_serial / 1e6, 3), "unit": "M paired-reads/s", "s": round(t_serial, 3),
                       "phase_ms": {"count_A": mean(serial, 3), "scan_B": mean(serial, 4), "vote_C": mean(serial, 5)}},
            "pipelined": {"value": round(n_samples * pairs / t_piped / 1e6, 3), "unit": "M paired-reads/s", "s": round(t_piped, 3),
                          "phase_ms": {"count_A": mean(piped, 3), "scan_B": mean(piped, 4), "vote_C": mean(piped, 5)}},
            "gain": round(t_serial / t_piped, 3), "identical_per_sample_results": bool(same),
            "what": "two contexts on one GPU, the reference resident as packed bases in both; a sample's phase A runs beside the other context's phases B-C "
                    "(two host threads, one lock per kind of phase); phase times are HIP events on each context's stream, so under overlap they include the slowdown by the neighbour"}


def dry_run(args, rank, world, local):
    """CPU self-test of the N > 1 plumbing: the launcher, the process group (gloo) and every exchange of localhgt_amd/dist.py
    on host tensors through the test adapter.  Measures nothing."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dist_cpu import FakeEngine, NumpyAdapter, unpack
    from localhgt_amd.dist import Exchange
    for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
        os.environ.setdefault(key, val)
    ex = Exchange.from_env(backend="gloo", adapter=NumpyAdapter())
    rng = np.random.default_rng(100 + rank)
    table = rng.choice(4, size=(1 << 16) * 4, p=[.6, .2, .1, .1]).astype(np.uint8)
    eng = FakeEngine(table, rng.integers(0, 300, size=1000))
    t0 = time.time()
    ex.merge_counts(eng)
    ex.sum_votes(eng)
    eng.rank, eng.n_new = rank, 3 + rank
    total = ex.sharded_scan(eng, 0.1, 0.08, 1000)
    mine = torch.tensor([int(unpack(eng.table.numpy()).sum()), int(eng.votes.sum())], dtype=torch.int64)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(allv, mine)
    ok = all(bool((v == allv[0]).all()) for v in allv) and total == sum(3 + r for r in range(world))
    ex.barrier()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "value": None, "n_gpus": world, "world_size": torch.distributed.get_world_size(),
                          "backend": "gloo", "exchanges_consistent": ok, "seconds": round(time.time() - t0, 3)}), flush=True)
    ex.close()
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
