#!/usr/bin/env python3
"""bench.py -- M paired-reads/s through the whole sketch->peak path (phases A->D), index and
packed reads resident in HBM, on N GPUs of one node (one process per GPU, RCCL over xGMI).

A step = counts_clear -> count_kmers (A) -> [count-table exchange] -> ref_scan (B) -> vote (C)
-> [vote all-reduce] -> write_intervals (D) over the synthetic workload of BASELINE.json
configs[1]: 1 Gbase reference (1000 x 1 Mbp), 10 M 150 bp pairs PER GPU (weak scaling: read shards
are independent), k=32 e=3, every read kept (--sample 1).  Prints ONE JSON line on rank 0.
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
    ap.add_argument("--pairs", type=int, default=10_000_000, help="read pairs per GPU")
    ap.add_argument("--contigs", type=int, default=1000)
    ap.add_argument("--contig-len", type=int, default=1_000_000)
    ap.add_argument("-k", type=int, default=32)
    ap.add_argument("-e", type=int, default=3)
    ap.add_argument("--shard-index", action="store_true", help="reference-sharded phase B: each rank holds 1/N of the index")
    ap.add_argument("--count-mode", type=int, default=-1, help="-1 = engine default (adaptive), 0 = direct CAS kernel, 1 = radix partition")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL exchange code even at world size 1 (self-test of the N>1 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=150_000)
    args = ap.parse_args()

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
    t0 = time.time()
    shard_index = args.shard_index and dist is not None
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
    ms = [0.0, 0.0, 0.0]
    for _ in range(args.steps):
        n_peaks, nf = step()
        for ph in range(3):
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
        tpath = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        default_workload = (args.contigs, args.contig_len, args.pairs, k, e) == (1000, 1_000_000, 10_000_000, 32, 3)
        if default_workload and os.path.exists(tpath):   # the PMC profile was taken on exactly this workload
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = {}
        vote_ach, vote_frac = roof(per["vote_C"], algo)
        count_ach, count_frac = roof(per["count_A"], algo)
        ref_bytes = args.contigs * args.contig_len * (4 * e + 64 * e)     # 204 B per reference base
        scan_ach, scan_frac = roof(per["scan_B"], ref_bytes)
        line = {
            "metric": "M paired-reads/s k-mer sketch->peak, UHGG-scale ref; %HBM roofline @1/2/4/8 GPU",
            "value": round(total_pairs / dt / 1e6, 4), "unit": "M paired-reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.contigs}x{args.contig_len} bp synthetic ref ({args.contigs * args.contig_len / 1e9:.2f} Gbase, "
                                   f"index resident), {args.pairs} 150bp pairs per GPU, k={k} e={e}, sample=1, phases A-D",
                       "pairs_per_gpu": args.pairs, "ref_bases": args.contigs * args.contig_len, "k": k, "e": e,
                       "parallelism": f"reads sharded x{world}" + (", index sharded" if shard_index else ", phase B replicated" if world > 1 else "")},
            "phase_ms": {"count_A": round(ms[0] / args.steps, 3), "scan_B": round(ms[1] / args.steps, 3), "vote_C": round(ms[2] / args.steps, 3)},
            "raw_peaks": n_peaks, "filtered_peaks": nf, "setup_s": round(setup_s, 2),
            "roofline": {"bound": "hbm", "kernel": "vote_kernel (phase C read re-scan: 714 probes/pair into peak_kmer; the dominant kernel)",
                         "achieved": vote_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": vote_frac,
                         "traffic": traffic.get("vote_kernel"), "algorithmic_bytes_per_launch": algo, "launch_ms": round(per["vote_C"], 3)},
            "roofline_other": {
                "count_A (part_hist+part_scatter_reads+part_scatter_keys+part_apply, all launches of one step)": {
                    "achieved": count_ach, "frac": count_frac, "algorithmic_bytes": algo, "ms": round(per["count_A"], 3),
                    "traffic": traffic.get("count_A"),
                    "note": "radix partition + LDS apply moves ~16 B/key of streaming traffic instead of one 64 B sector per key, so the sector-model fraction can exceed the random-access ceiling"},
                "scan_B (ref_flags+window_peak+interval_mask+tile_scan+register_peaks)": {
                    "achieved": scan_ach, "frac": scan_frac, "algorithmic_bytes": ref_bytes, "ms": round(per["scan_B"], 3),
                    "traffic": traffic.get("scan_B")}},
        }
        if world == 1 and not args.no_cpu_baseline:
            eng.pairs_clear()
            try:
                line["cpu_baseline"] = cpu_baseline(k, e, 20, 1_000_000, args.cpu_pairs, 1, 2, lambda: Engine(k, e, device=local))
            except Exception as ex:  # the baseline must never sink the measurement
                line["cpu_baseline"] = {"error": str(ex)}
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
