#!/usr/bin/env python3
"""bench.py -- M paired-reads/s through the whole sketch->peak path (phases A->D), index and
packed reads resident in HBM, on N GPUs of one node (one process per GPU, RCCL over xGMI).

A step = counts_clear -> count_kmers (A) -> [count-table exchange] -> ref_scan (B) -> vote (C)
-> [vote all-reduce] -> write_intervals (D) over a synthetic workload of BASELINE.json.  Default =
configs[2], the configuration the metric is quoted on ("UHGG-scale ref") and the largest that fits one
GPU: 13 Gbase reference (13000 x 1 Mbp, 156 GB of index resident in HBM), 100 M 150 bp pairs PER GPU
(weak scaling: read shards are independent), k=32 e=3, every read kept (--sample 1).
`--workload 1g` = configs[1] (1 Gbase, 10 M pairs).

Output.  Rank 0 prints ONE compact JSON line (< 4 KB: tools/benchlib/compact.py) right after the timed headline and again --
with the later fields filled in -- after every further stage; the LAST line on stdout is the record.  Everything else (full
roofline entries, every secondary leg, the PMC sums) goes to bench_detail.json next to this script (--detail-out to move it).
`python3 bench.py --gpus N` is self-contained: without WORLD_SIZE in the environment it starts its N ranks itself.

What the line holds besides the contract's fields (all of it measured inside this run):
  roofline      the dominant kernel among phase A's family, ref_flags* and the vote kernel: launch time by HIP events on the engine's
                stream; `traffic` from rocprofv3 --pmc child runs of this same command.  frac = (2 x FETCH_SIZE + WRITE_SIZE) / time /
                8 TB/s (every fabric read request of these kernels is a 128-B line fill tallied at 64 B), frac_raw = the counters as
                they are, frac_model = SURVEY 8d's one-sector-per-reference-probe bytes (model_exceeded where the code avoids those
                probes), frac_needed = the bytes the algorithm AS BUILT must move (its own key and probe counts, lhgt_work_stats) and
                overfetch = counter bytes / needed bytes  (tools/benchlib/roofline.py)
  value_found   the same read count (100 M pairs) drawn from 300 genomes of the same 13 Gbase reference: the regime in which the path
                finds its planted transfers (the headline's half-of-everything sample saturates the 2^32-slot table and votes nothing)
  cpu_baseline  the CPU restatement on all host cores (kind "port": the checker, identical_to_gpu) and `reference`: the compiled
                reference itself (oracle/_ref/extract_ref_raw, -t 10) run in the background on the same files
  secondary     one number per other regime: configs[1], the SNP 1 % sample, the CLI's default --sample 2e9, a ragged catalogue,
                the packed reference, configs[4]-scale on one GPU (k = 32 / 21), and pairs/s from FASTQ files (e2e: last of all, in a
                child process of its own -- 23 GB of files through the whole host pipeline must not be able to take the line along)
  at N > 1      value = the REPLICATED form (every rank scans the whole reference: per-GPU work fixed, which is what "weak" means);
                sharded_index = the same step with phase B sharded over the ranks; exchange_ms; n1_equivalent_ms
"""
import argparse
import faulthandler
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from benchlib import METRIC                                                             # noqa: E402
from benchlib.compact import compact_line, emit                                          # noqa: E402,F401
from benchlib.files import synth_files, synth_files_sliced, write_fasta, write_fastq    # noqa: E402,F401  (tests import them from here)
from benchlib.launch import dry_run, self_launch                                         # noqa: E402
from benchlib.plan import HBM_BYTES, check_fits, memory_plan                             # noqa: E402
from benchlib.pmc import (KERNEL_SOURCES, PMC_PASS_L2, PMC_PASSES, collect_pmc, committed_traffic, pmc_traffic,  # noqa: E402,F401
                          source_stamp)
from benchlib.recall import interval_recall, planted_breakpoints                        # noqa: E402,F401
from benchlib.roofline import HBM_PEAK_GBS, rooflines                                    # noqa: E402

HEADLINE = (13000, 100_000_000, 0, False, 0)    # contigs, pairs, sample_contigs, ragged, snp


def parse_args(argv=None):
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
    ap.add_argument("--snp", type=int, default=0, help="SNPs per 1000 bases of the sample genomes (10 = the reference's species20_snp0.01 model)")
    ap.add_argument("--ragged", action="store_true", help="the same base stream cut into ~118 k contigs of a catalogue-like length distribution (localhgt_amd.synth.ragged_cuts)")
    ap.add_argument("-k", type=int, default=32)
    ap.add_argument("-e", type=int, default=3)
    ap.add_argument("--shard-index", action="store_true", help="N > 1: time ONLY the reference-sharded phase B (each rank holds 1/N of the index)")
    ap.add_argument("--replicate-index", action="store_true", help="N > 1: time ONLY the replicated form (the whole index on every GPU, no exchange in phase B)")
    ap.add_argument("--ref-form", choices=["index", "packed"], default="packed",
                    help="resident form of the reference: the packed bases (3/8 B/base), phase B recomputing the hashes, with the slot list -- the "
                         "reference's k-mer positions by hash bucket, 6 B/base -- next to them (default since round 5); or the index file's hashes "
                         "(12 B/base at e=3: rounds 1-4's headline, now the `index_form` leg of the same line)")
    ap.add_argument("--no-slot-list", action="store_true", help="packed form without the slot list: phase B's position-ordered kernels (A/B)")
    ap.add_argument("--count-mode", type=int, default=-1, help="-1 = engine default (adaptive), 0 = direct CAS kernel, 1 = radix partition")
    ap.add_argument("--debug", type=int, default=0, help="engine debug/A-B switches (include/localhgt_hip.h: lhgt_set_debug)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL exchange code even at world size 1 (self-test of the N>1 path)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: the exchanges staged through host memory (localhgt_amd/dist.py) -- for ranks that share a GPU (rehearsal of the N > 1 path on a one-GPU box) and for --dry-run")
    ap.add_argument("--dry-run", action="store_true", help="launcher + process group + exchanges on host tensors, no GPU and no measurement (CPU self-test of the N>1 plumbing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=400_000)
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads and the from-FASTQ leg")
    ap.add_argument("--full", action="store_true", help="also the slow extras: PMC of the ragged leg, the packed e2e legs, two samples side by side")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect HBM traffic with rocprofv3 --pmc child runs")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed picked-forms-vs-exact-forms check")
    ap.add_argument("--no-stats", action="store_true", help="skip the untimed step that counts keys and probes (the PMC children: their counters must see the timed step alone)")
    ap.add_argument("--pmc-out", default=None, help="also write the PMC summary of the headline to this JSON file")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes")
    ap.add_argument("--quiet", action="store_true", help="(PMC children) no detail file")
    ap.add_argument("--make-cpu-files", default=None, metavar="DIR", help="(internal) write the CPU legs' FASTA/FASTQ/index into DIR and exit")
    ap.add_argument("--e2e-child", default=None, metavar="JSON", help="(internal) run the from-FASTQ legs and write their record to JSON")
    ap.add_argument("--device", type=int, default=0, help="(internal, with --e2e-child) the GPU to use")
    args = ap.parse_args(argv)
    wl_contigs, wl_pairs = (13000, 100_000_000) if args.workload == "uhgg" else (1000, 10_000_000)
    args.contigs = args.contigs or wl_contigs
    args.pairs = args.pairs or wl_pairs
    return args


def workload_key(args):
    return (args.contigs, args.pairs, args.sample_contigs, args.ragged, args.snp)


def main():
    faulthandler.enable()        # a native abort or fault leaves the Python frames it happened under on stderr
    args = parse_args()
    if args.make_cpu_files:
        from benchlib.cpu import make_files
        make_files(args.make_cpu_files, args.cpu_pairs)
        return
    if args.e2e_child:
        from benchlib.secondary import e2e_child
        e2e_child(args.e2e_child, args.k, args.e, args.device, args.full)
        return
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
    headline = workload_key(args) == HEADLINE and (args.k, args.e, args.contig_len) == (32, 3, 1_000_000)
    detail_path = None if (args.quiet or rank != 0) else args.detail_out
    workload_tag = (f"{args.contigs}x{args.contig_len}_{args.pairs}_k{k}_e{e}" + (f"_s{args.sample_contigs}" if args.sample_contigs else "")
                    + (f"_snp{args.snp}" if args.snp else "") + ("_packed" if args.ref_form == "packed" else "") + ("_ragged" if args.ragged else ""))
    extras = world == 1 and not args.no_extras and not args.force_dist and not args.debug

    # ---- does it fit?  (before anything is allocated; DESIGN.md 6 holds the configs[3] plan)
    shard_only = args.shard_index and world > 1
    use_list = args.ref_form == "packed" and not args.no_slot_list and not shard_only
    plan = check_fits(memory_plan(args.pairs, args.contigs * args.contig_len, args.contigs, k, e, L, args.ref_form == "packed", world, shard_only, slot_list=use_list),
                      HBM_BYTES, f"--gpus {world}: {args.pairs} pairs per GPU vs {args.contigs * args.contig_len / 1e9:.1f} Gbase "
                                 f"({'packed' if args.ref_form == 'packed' else 'index'} form{', sharded' if shard_only else ''})")

    # ---- N > 1: the process group first.  Rank 0's store takes the rendezvous port within its first seconds (the launcher only
    # picked a free one); the PMC children below then run while the other ranks wait at the first barrier
    dist = None
    if world > 1 or args.force_dist:
        from localhgt_amd.dist import Exchange
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(key, val)
        dist = Exchange.from_env(backend=args.backend)

    # ---- the compiled reference on host cores, in the background from the very beginning (rank 0, N = 1)
    cpu_tmp, ref_run, cpu_files = None, None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_dist:
        from benchlib import cpu as cpu_legs
        try:
            cpu_tmp = tempfile.TemporaryDirectory(prefix="lhgt_cpu_")
            cpu_files = cpu_legs.make_files_in_child(cpu_tmp.name, args.cpu_pairs)
            if os.path.exists(cpu_legs.REF_BIN):
                ref_run = cpu_legs.ReferenceRun(*cpu_files, cpu_tmp.name, args.cpu_pairs, threads=10)
        except Exception as ex:   # the baseline must never sink the measurement
            cpu_files = None
            print(f"bench: cpu files: {ex}", file=sys.stderr)

    # ---- measured HBM traffic of the headline: child runs of this command under rocprofv3 --pmc, while this process holds nothing
    # on the GPU.  At N > 1 rank 0 does the same with N = 1 children (every rank runs phases A and C on a read shard of the N = 1
    # size and shape, and the replicated phase B on the whole reference: its kernels are the N = 1 kernels)
    pmc, pmc_note, traffic, traffic_src = None, "", {}, None
    if rank == 0 and not args.no_pmc and not args.force_dist:
        t0 = time.time()
        pmc, pmc_note = collect_pmc(args, PMC_PASSES + ([PMC_PASS_L2] if world == 1 else []))
        if pmc:
            traffic = pmc_traffic(pmc)
            traffic_src = (f"rocprofv3 --pmc passes of this run ({time.time() - t0:.0f} s, 1 step each; FETCH_SIZE+TCP_TCC_READ_REQ, WRITE_SIZE+TCC_EA0_RDREQ, TCC_HIT+TCC_MISS)"
                           + (", as N = 1 children of rank 0 on this rank's per-GPU workload" if world > 1 else ""))
            if args.pmc_out:
                json.dump({"tag": workload_tag, "kernels": pmc, "per_step": traffic,
                           "_stamp": {ph: source_stamp(s) for ph, s in KERNEL_SOURCES.items()}}, open(args.pmc_out, "w"), indent=1, sort_keys=True)
    if rank == 0:
        fresh, _stale = committed_traffic(workload_tag)
        for ph, rec in fresh.items():
            if ph not in traffic:
                traffic[ph] = rec
                traffic_src = traffic_src or "profiles/traffic_per_launch.json (measured on these sources; the live rocprofv3 passes failed)"

    import torch
    from localhgt_amd.engine import Engine
    from benchlib.legs import Workload, slot_list_streamed_bytes, verify_forms
    local = local % max(1, torch.cuda.device_count())          # --backend gloo: ranks may share a GPU
    torch.cuda.set_device(local)

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
    if args.ref_form == "packed":
        # the slot list of the resident reference (include/localhgt_hip.h: lhgt_slot_list): built before the first scan that has a
        # use for it -- untimed, like the reference load it belongs to -- instead of before the second (the engine's own rule)
        big_ref = args.contigs * args.contig_len >= 1 << 32      # below that the engine's own rule leaves the list alone
        eng.slot_list(2 if use_list and not forms[0] and big_ref else 1 if use_list else 0)
    eng.synth_options(args.snp, 20, args.sample_contigs)
    eng.synth_pairs(1, 2, args.contigs, args.contig_len, rank * args.pairs, args.pairs, L)   # this rank's shard, packed, resident
    eng.synchronize()
    setup_s = time.time() - t0
    wl = Workload(eng, dist, rank, world, forms[0], out_path)

    verify, stats = None, None
    if world == 1 and not args.debug and not args.force_dist and not args.no_stats:
        stats = wl.stats_step()                              # untimed: the run's own key and probe counts (needed-bytes model)
        if args.ref_form == "packed" and eng.slot_list()["bytes"]:
            stats["slot_list_bytes"] = slot_list_streamed_bytes(eng)
        if not args.no_verify:
            verify = verify_forms(eng)
    dt, per_ms, n_peaks, nf = wl.run(args.steps, args.warmup)
    scan = eng.scan_info()
    vote_form = eng.vote_info()
    xch_main = dict(wl.xch)
    moved_main = dict(dist.moved) if dist else None
    sl_info = eng.slot_list() if args.ref_form == "packed" else {"entries": 0, "bytes": 0}      # of the form the timed steps ran on
    sl_cost = None
    if sl_info["entries"] and world == 1 and not args.debug and not args.no_stats and scan["form"] in ("slot-first", "slot-single"):   # (not in the PMC children: their counters must see the timed step alone)
        # the list is a per-reference precompute built OUTSIDE the timed steps (like the reference load): what it cost, the same steps
        # without it (debug bit 25: phase B's position-ordered kernels on the same packed reference), and after how many samples of one
        # resident reference it has paid for itself.  `bin/extract_ref --batch` is the shipped entry point that reaches this state.
        eng.set_debug(1 << 25)
        dt0, per0, n0, nf0 = wl.run(2, 0)
        eng.set_debug(0)
        ms0, ms1 = dt0 / 2 * 1e3, dt / args.steps * 1e3
        sl_cost = {"slot_list_build_ms": sl_info.get("build_ms"), "slot_list_GB": round(sl_info["bytes"] / 1e9, 1),
                   "value_without_list": round(args.pairs * 2 / dt0 / 1e6, 3), "ms_per_step_without_list": round(ms0, 2),
                   "scan_B_ms_without_list": round(per0[1], 2), "same_peaks": (n0, nf0) == (n_peaks, nf),
                   "break_even_samples": round(sl_info.get("build_ms", 0.0) / (ms0 - ms1), 1) if ms0 > ms1 else None,
                   "note": "value is measured with the reference AND its slot list resident (the list is built once per resident reference, untimed); value_without_list = "
                           "the same steps on phase B's position-ordered kernels; break_even_samples = build time / time saved per sample.  One process per sample "
                           "(scripts/pipeline.sh:35) never reaches the list; `extract_ref --batch MANIFEST` does (secondary.batch_13g_from_files, from FASTQ files)"}
    other_form = None
    if len(forms) > 1:                                   # the other form of phase B, a few steps, same reads
        if args.ref_form == "packed":
            eng.slot_list(2 if use_list and not forms[1] and big_ref else 1 if use_list else 0)   # a shard of 1/N of the reference: the engine's own rule (lists from 2^32 positions on)
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
    ref_bases = args.contigs * args.contig_len
    n_contigs_here = args.contigs
    if forms[0]:
        n_contigs_here = args.contigs * (rank + 1) // world - args.contigs * rank // world
        ref_bases = n_contigs_here * args.contig_len          # this rank's contig range
        if world > 1 and traffic.get("ref_flags"):
            share = ref_bases / (args.contigs * args.contig_len)
            traffic["ref_flags"] = {kk: (int(v * share) if isinstance(v, (int, float)) else v) for kk, v in traffic["ref_flags"].items()}
    if args.ragged:
        from localhgt_amd.synth import ragged_cuts
        n_contigs_here = len(ragged_cuts(args.contigs * args.contig_len)) - 1
    roof, dominant = rooflines(k, e, L, args.pairs, ref_bases, n_contigs_here, args.ref_form == "packed", per_ms, scan, n_peaks, traffic, traffic_src, stats, vote_form)
    # bytes a step cannot avoid: the packed reads twice (A and C), the resident index once, count table written and read,
    # peak_kmer cleared (E:1458) -- everything else is the price of random access
    read_store = args.pairs * 2 * 3 * ((L + 31) // 32 + 1) * 4
    compulsory = 2 * read_store + (ref_bases * 3 // 8 if args.ref_form == "packed" else ref_bases * 4 * e) + 2 * ((1 << k) // 4) + (1 << k) * 4
    step_s = dt / args.steps
    xch_ms = {kk: round(v / args.steps * 1e3, 3) for kk, v in xch_main.items()} if dist else None
    xch_bytes = None
    if dist:                     # what rank 0 sent + received per step in each exchange, and at what rate (host clock around the call: device work of the merge included)
        xch_bytes = {kk: {"bytes_per_step": int(moved_main[kk] // args.steps), "GB_per_s": round(moved_main[kk] / max(xch_main[kk], 1e-9) / 1e9, 2) if xch_main[kk] > 0 else None}
                     for kk in moved_main}
        print("bench: exchanges of rank 0 per step (" + dist.backend + "): " + "; ".join(
            f"{kk} {v['bytes_per_step'] / 1e6:.1f} MB in {xch_ms[kk]:.2f} ms" + (f" = {v['GB_per_s']} GB/s" if v["GB_per_s"] else "") for kk, v in xch_bytes.items()),
            file=sys.stderr, flush=True)
    resident_txt = ("index resident" if args.ref_form != "packed" else
                    f"resident as packed bases + slot list = its k-mer positions by hash bucket, {(sl_info['bytes'] + ref_bases * 3 // 8) / 1e9:.0f} GB, list built once per resident reference outside the timed steps: slot_list_build_ms; reached by extract_ref --batch" if sl_info["entries"] else "packed bases resident")
    cfg_no = 2 if headline else 1 if workload_key(args) == (1000, 10_000_000, 0, False, 0) else "-"
    sample_txt = (f"drawn from {args.sample_contigs} of its contigs" if args.sample_contigs else "drawn from half of its contigs") + (f", SNP {args.snp / 10:g} %" if args.snp else "")
    detail = {
        "metric": METRIC, "value": round(total_pairs / dt / 1e6, 4), "unit": "M paired-reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{cfg_no}]: {args.contigs}x{args.contig_len} bp synthetic ref ({args.contigs * args.contig_len / 1e9:.2f} Gbase"
                               f"{', cut into a ragged catalogue' if args.ragged else ''}, {resident_txt}), "
                               f"{args.pairs} 150bp pairs per GPU {sample_txt}, k={k} e={e}, sample=1, phases A-D",
                   "pairs_per_gpu": args.pairs, "ref_bases": args.contigs * args.contig_len, "k": k, "e": e,
                   "parallelism": f"reads sharded x{world}" + (", index sharded" if forms[0] else ", phase B replicated on every GPU (per-GPU work fixed)" if world > 1 else "")},
        "world_size": torch.distributed.get_world_size() if dist else 1, "backend": dist.backend if dist else None,
        "phase_ms": {"count_A": round(per_ms[0], 3), "scan_B": round(per_ms[1], 3), "vote_C": round(per_ms[2], 3)},
        "exchange_ms": xch_ms, "exchange_bytes": xch_bytes,
        "n1_equivalent_ms": round(step_s * 1e3 - sum(xch_ms.values()), 3) if xch_ms else round(step_s * 1e3, 3),
        "sharded_index" if (other_form and forms[1]) else "replicated_index": other_form,
        "scan_B_form": scan, "vote_form": vote_form, "work_stats": stats, "memory_plan_bytes": plan, "slot_list": sl_info, "slot_list_cost": sl_cost,
        "raw_peaks": n_peaks, "filtered_peaks": nf, "setup_s": round(setup_s, 2),
        "planted_transfers": interval_recall(out_path, planted_breakpoints(args.contigs, args.contig_len, args.sample_contigs)) if world == 1 and not args.ragged else None,
        "verify": verify,
        "roofline": roof[dominant],
        "roofline_other": {ph: roof[ph] for ph in roof if ph != dominant},
        "dominant": dominant,
        "compulsory": {"bytes_per_step": compulsory, "frac_of_peak": round(compulsory / step_s / (HBM_PEAK_GBS * 1e9), 4),
                       "what": "packed reads twice + resident index once + count table written and read + peak_kmer cleared; a step at HBM peak would take "
                               f"{compulsory / (HBM_PEAK_GBS * 1e9) * 1e3:.0f} ms"},
        "pmc_kernels": pmc, "pmc_note": pmc_note or None,
        "note": "roofline: frac = (2 x FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s (a fabric read request on gfx950 is a 128-B line fill tallied at 64 B: "
                "tools/probe_shapes.hip, profiles/r02/); frac_raw = counters as they are; frac_model = SURVEY 8d's 64-B-sector-per-reference-probe bytes "
                "(model_exceeded where partition, L2 bitmap or lite scan avoid the probes); frac_needed = bytes the algorithm as built must move "
                "(bytes_needed, needed_is: its own key / probe counts x 128-B lines + streams) and overfetch = counter bytes / needed bytes; "
                "tools/benchlib/roofline.py, DESIGN.md 5",
    }
    emit(detail, "headline", detail_path)                    # the metric is out: nothing below can lose it

    eng.pairs_clear()
    if extras:
        from benchlib import secondary
        try:
            secondary.run_all(detail, eng, args, wl, local, lambda stage: emit(detail, stage, detail_path))
        except Exception as ex:
            detail.setdefault("secondary", {})["error"] = str(ex)[:200]
    else:
        eng.close()
    if dist:
        dist.close()
    if cpu_files:
        from benchlib import cpu as cpu_legs
        try:
            ref_rec = ref_run.join() if ref_run else None
            detail["cpu_baseline"] = cpu_legs.port_and_gpu(*cpu_files, cpu_tmp.name, args.cpu_pairs, local, reference=ref_rec)
        except Exception as ex:  # the baseline must never sink the measurement
            detail["cpu_baseline"] = {"error": str(ex)[:200]}
    if cpu_tmp:
        cpu_tmp.cleanup()
    if extras and rank == 0 and world == 1:
        from benchlib import secondary
        emit(detail, "cpu baseline", detail_path)
        from localhgt_amd.engine import pool_trim
        pool_trim()                  # the device blocks this process kept from its closed engines: the child gets the whole GPU
        secondary.run_e2e(detail, args, local, lambda stage: emit(detail, stage, detail_path))
    emit(detail, "final", detail_path)
    try:
        os.remove(out_path)
    except OSError:
        pass
    if isinstance(detail.get("cpu_baseline"), dict) and detail["cpu_baseline"].get("identical_to_gpu") is False:
        sys.exit("bench: the GPU path and the CPU baseline wrote different interval files for the same inputs")


if __name__ == "__main__":
    main()
