#!/usr/bin/env python3
"""bench.py -- M paired-reads/s through the whole sketch->peak path (phases A->D), index and
packed reads resident in HBM, on N GPUs of one node (one process per GPU, RCCL over xGMI).

A step = counts_clear -> count_kmers (A) -> [count-table exchange] -> ref_scan (B) -> vote (C)
-> [vote all-reduce] -> write_intervals (D) over a synthetic workload of BASELINE.json.  Default =
configs[2], the configuration the metric is quoted on ("UHGG-scale ref") and the largest that fits one
GPU: 13 Gbase reference (13000 x 1 Mbp, 156 GB of index resident in HBM), 100 M 150 bp pairs PER GPU
(weak scaling: read shards are independent), k=32 e=3, every read kept (--sample 1).
`--workload 1g` = configs[1] (1 Gbase, 10 M pairs).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PAIR = lambda L, k, e: 2 * (L - k + 1) * e * 64 + (2 * L + 3) // 4   # one 64 B sector per probe + packed bases
HBM_PEAK_GBS = 8000.0                                                               # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(k, e, n_contigs, contig_len, n_pairs, seed_ref, seed_reads, eng_factory):
    """Time the CPU oracle (oracle/lhgt_oracle.c, all host cores) on a bounded sample of the same
    synthetic workload.  The oracle is the checker/baseline here, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_api
    from conftest import build_oracle
    orc = oracle_api.Oracle(build_oracle())
    orc.set_pretouch(True)     # table page faults before the phase timers, like the reference's memsets (E:1416, 1458)
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory(prefix="lhgt_cpu_") as tmp:
        with eng_factory() as eng:
            eng.rng_seed(1)
            eng.coder_generate()
            ref = eng.synth_reference(seed_ref, n_contigs, contig_len, want_host=True)
            m1, m2 = eng.synth_pairs(seed_ref, seed_reads, n_contigs, contig_len, 0, n_pairs, 150, want_host=True)
        fa, f1, f2 = (os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq"))
        with open(fa, "wb") as f:
            for c in range(n_contigs):
                f.write(b">g%d\n" % (c + 1))
                f.write(ref[c * contig_len:(c + 1) * contig_len].tobytes())
                f.write(b"\n")
        for path, m, suf in ((f1, m1, b"1"), (f2, m2, b"2")):
            a = m.reshape(n_pairs, 150)
            qual = b"I" * 150
            with open(path, "wb") as f:
                for i in range(n_pairs):
                    f.write(b"@r%09d/%s\n" % (i, suf) + a[i].tobytes() + b"\n+\n" + qual + b"\n")
        rc, rep = orc.run(f1, f2, fa, os.path.join(tmp, "interval.txt"), 0.1, 0.08, cores, k, 3000000, e, 1, 1.0)
        if rc != 0:
            return None
        t = rep.t_count + rep.t_scan + rep.t_vote
        return {"value": round(n_pairs / t / 1e6, 6), "unit": "M paired-reads/s", "cores": cores, "kind": "port",
                "sample": f"{n_pairs} pairs x 150 bp vs {n_contigs} x {contig_len} bp synthetic contigs, k={k} e={e}, "
                          f"phases A+B+C of oracle/lhgt_oracle.c (index build excluded): "
                          f"A {rep.t_count:.2f}s B {rep.t_scan:.2f}s C {rep.t_vote:.2f}s",
                "raw_peaks": int(rep.n_peaks)}


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
    ap.add_argument("-k", type=int, default=32)
    ap.add_argument("-e", type=int, default=3)
    ap.add_argument("--shard-index", action="store_true", help="reference-sharded phase B: each rank holds 1/N of the index (the default at N > 1)")
    ap.add_argument("--replicate-index", action="store_true", help="at N > 1 keep the whole index on every GPU and scan it redundantly (no exchange in phase B)")
    ap.add_argument("--count-mode", type=int, default=-1, help="-1 = engine default (adaptive), 0 = direct CAS kernel, 1 = radix partition")
    ap.add_argument("--debug", type=int, default=0, help="engine debug/A-B switches (include/localhgt_hip.h: lhgt_set_debug)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL exchange code even at world size 1 (self-test of the N>1 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=400_000)
    args = ap.parse_args()
    wl_contigs, wl_pairs = (13000, 100_000_000) if args.workload == "uhgg" else (1000, 10_000_000)
    args.contigs = args.contigs or wl_contigs
    args.pairs = args.pairs or wl_pairs

    import torch
    from localhgt_amd.engine import Engine
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        from localhgt_amd.dist import Exchange
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(key, val)
        dist = Exchange.from_env(backend="nccl")

    k, e, L = args.k, args.e, 150
    eng = Engine(k, e, device=local)
    eng.rng_seed(1)
    eng.coder_generate()
    if args.count_mode >= 0:
        eng.set_count_mode(args.count_mode)
    if args.debug:
        eng.set_debug(args.debug)
    t0 = time.time()
    # phase B shards by contig range -- each rank scans 1/N of the reference and the peaks are exchanged
    shard_index = dist is not None and (args.shard_index or (world > 1 and not args.replicate_index))
    if shard_index:
        eng.synth_reference_shard(1, args.contigs, args.contig_len, rank, world)  # this rank's contig range only
    else:
        eng.synth_reference(1, args.contigs, args.contig_len)                   # whole index resident in HBM
    eng.synth_pairs(1, 2, args.contigs, args.contig_len, rank * args.pairs, args.pairs, L)   # this rank's shard, packed, resident
    eng.synchronize()
    setup_s = time.time() - t0
    out_path = os.path.join(tempfile.gettempdir(), f"lhgt_bench_interval_{os.getpid()}.txt")

    def step():
        eng.counts_clear()
        eng.count_kmers()
        if dist:
            dist.merge_counts(eng)
        if shard_index:
            n_peaks = dist.sharded_scan(eng, 0.1, 0.08, 300_000_000)
        else:
            n_peaks = eng.ref_scan(0.1, 0.08, 300_000_000)
        eng.vote()
        if dist:
            dist.sum_votes(eng)
        nf = eng.write_intervals(out_path) if rank == 0 else -1
        return n_peaks, nf

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.time()
    ms = [0.0, 0.0, 0.0, 0.0]
    for _ in range(args.steps):
        n_peaks, nf = step()
        for ph in range(4):
            ms[ph] += eng.phase_ms(ph)
    fence()
    dt = time.time() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        total_pairs = args.pairs * world * args.steps
        algo = ALGO_BYTES_PER_PAIR(L, k, e) * args.pairs    # algorithmic bytes of one scan launch over this GPU's pairs
        per = {"count_A": ms[0] / args.steps, "scan_B": ms[1] / args.steps, "vote_C": ms[2] / args.steps}   # HIP events on the engine stream

        def roof(ms_launch, bytes_launch):
            ach = bytes_launch / (ms_launch * 1e-3) / 1e9
            return round(ach, 2), round(ach / HBM_PEAK_GBS, 4)

        # measured HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc_collect.sh;
        # FETCH_SIZE of random 4-byte probes = TCC_EA0_RDREQ x 64 B, the streaming kernels' FETCH_SIZE doubled per
        # MI355X_MICROARCH.md section HBM); committed with the profile it came from
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_per_launch.json")   # {workload tag: {phase: bytes}}
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = {}
        ref_bytes = args.contigs * args.contig_len * (4 * e + 64 * e)     # 204 B per reference base
        if shard_index:
            ref_bytes = (args.contigs * (rank + 1) // world - args.contigs * rank // world) * args.contig_len * (4 * e + 64 * e)   # this rank's contig range
        # candidates for "the dominant kernel": single kernels timed by their own HIP events (ref_flags: one launch per step;
        # vote_kernel: one launch per resident batch of <= 16 Mi pairs, phase C holds nothing else), and phase A's kernel family
        n_batches = -(-args.pairs // (16 << 20))
        scan = eng.scan_info()
        kern = {"count_A": per["count_A"], "ref_flags": ms[3] / args.steps, "vote_C": per["vote_C"]}
        phases = {
            "count_A": ("phase A kernel family (part_scatter_reads + part_scatter_keys + part_apply per <= 4 Mi-pair chunk; "
                        "count_direct below k = 26): 714 table updates per pair", algo, None),
            "ref_flags": (("ref_flags_lite (phase B on a nearly saturated table: one probe per base until a hash reads 3, all e at every 8th base; "
                           "algorithmic bytes as for the exact form)" if scan["lite"] else
                           "ref_flags (phase B: e random 2-bit table probes + e index words per reference base)") + ", 1 launch per step", ref_bytes, 1),
            "vote_C": (f"vote_kernel (phase C read re-scan: 714 probes per pair into peak_kmer), {n_batches} launches per step", algo, n_batches),
        }
        # the dominant KERNEL: of the two that are one kernel each (phase A is a family of three kernels per chunk, reported below)
        dominant = max(("ref_flags", "vote_C"), key=kern.get)
        dom_ach, dom_frac = roof(kern[dominant], phases[dominant][1])
        tkey = {"count_A": "count_A", "ref_flags": "ref_flags", "vote_C": "vote_kernel"}
        workload_tag = f"{args.contigs}x{args.contig_len}_{args.pairs}_k{k}_e{e}"
        tr = traffic.get(workload_tag, {}) if not (shard_index and world > 1) else {}   # measured on the single-GPU, whole-index run
        line = {
            "metric": "M paired-reads/s k-mer sketch->peak, UHGG-scale ref; %HBM roofline @1/2/4/8 GPU",
            "value": round(total_pairs / dt / 1e6, 4), "unit": "M paired-reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{2 if (args.contigs, args.pairs) == (13000, 100_000_000) else 1 if (args.contigs, args.pairs) == (1000, 10_000_000) else '-'}]: "
                                   f"{args.contigs}x{args.contig_len} bp synthetic ref ({args.contigs * args.contig_len / 1e9:.2f} Gbase, "
                                   f"index resident), {args.pairs} 150bp pairs per GPU, k={k} e={e}, sample=1, phases A-D",
                       "pairs_per_gpu": args.pairs, "ref_bases": args.contigs * args.contig_len, "k": k, "e": e,
                       "parallelism": f"reads sharded x{world}" + (", index sharded" if shard_index else ", phase B replicated" if world > 1 else "")},
            "phase_ms": {kk: round(v, 3) for kk, v in per.items()},
            "scan_B_form": scan,
            "raw_peaks": n_peaks, "filtered_peaks": nf, "setup_s": round(setup_s, 2),
            "roofline": {"bound": "hbm", "kernel": phases[dominant][0] + " -- the dominant kernel of this workload",
                         "achieved": dom_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom_frac,
                         "traffic": tr.get(tkey[dominant]), "algorithmic_bytes_per_step": phases[dominant][1],
                         "ms_per_step": round(kern[dominant], 3), "launches_per_step": phases[dominant][2],
                         "launch_ms": round(kern[dominant] / phases[dominant][2], 3) if phases[dominant][2] else None},
            "roofline_other": {ph: {"kernel": phases[ph][0], "achieved": roof(kern[ph], phases[ph][1])[0], "frac": roof(kern[ph], phases[ph][1])[1],
                                    "algorithmic_bytes_per_step": phases[ph][1], "ms_per_step": round(kern[ph], 3),
                                    "launches_per_step": phases[ph][2], "traffic": tr.get(tkey[ph])}
                               for ph in kern if ph != dominant},
            "note": "fractions use SURVEY 8d's sector model (one 64 B HBM sector per probe); the radix-partitioned count, the L2-resident vote "
                    "prefilter and the lite reference scan do the same work with far fewer HBM bytes (roofline.traffic is the measured figure), so "
                    "their fractions can exceed 1. The limits they actually run against are request rates measured by microbenchmark "
                    "(profiles/r01_probe_*): ~55 G random requests/s from HBM (ref_flags: 96 % of it), ~254 G/s from L2 (sparse vote: 80 % of it); DESIGN.md 4",
        }
        if world == 1 and not args.no_cpu_baseline:
            eng.pairs_clear()
            try:
                line["cpu_baseline"] = cpu_baseline(k, e, 20, 1_000_000, args.cpu_pairs, 1, 2, lambda: Engine(k, e, device=local))
            except Exception as ex:  # the baseline must never sink the measurement
                line["cpu_baseline"] = {"error": str(ex)}
        # RCCL writes its version banner through C stdio, which on a pipe is flushed at exit, i.e. after Python's output:
        # flush it now so that the JSON line is the last thing on stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    eng.close()
    if dist:
        dist.close()
    try:
        os.remove(out_path)
    except OSError:
        pass


if __name__ == "__main__":
    main()
