"""The timed loop of bench.py and the other regimes of the same path (N = 1), a few steps each."""
import os
import shutil
import tempfile
import time

from .files import near_gpu, synth_files, synth_files_sliced
from .recall import interval_recall, planted_breakpoints
from .roofline import rooflines

LIVE = "rocprofv3 --pmc passes of this run on this workload"


class Workload:
    """a step = counts_clear -> count_kmers (A) -> [count-table exchange] -> ref_scan (B) -> vote (C) -> [vote all-reduce] ->
    write_intervals (D)"""

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
        if self.dist:
            for key in self.dist.moved:
                self.dist.moved[key] = 0
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

    def stats_step(self):
        """one untimed step with the work counters on (lhgt_work_stats): keys routed by phase A, table probes of phase B's probe
        kernel, probes the vote sent on to the next filter level -- the inputs of the roofline's needed-bytes model"""
        self.eng.work_stats(1)
        self.step()
        self.eng.synchronize()
        return self.eng.work_stats(0)


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


def slot_list_streamed_bytes(engine):
    """the bytes of the slot list a scan streams: its ENTRIES (6 bytes each, 10 with the second hash) -- the regions they lie in hold
    7.5 % more since they are sized from a sampled histogram (lhgt_slot_list reports the regions)"""
    sl = engine.slot_list()
    if not sl["entries"]:
        return 0
    return (10 if sl["bytes"] > 8 * sl["entries"] else 6) * sl["entries"]


def leg(engine, out_path, k, e, pairs, n_contigs, contig_len, steps=3, sample_contigs=0, traffic=None, ref_bases=None, packed=False, recall=True,
        want_stats=True, L=150):
    """one secondary workload on a loaded engine: a stats step, a warm-up step, `steps` timed ones"""
    w = Workload(engine, None, 0, 1, False, out_path)
    stats = w.stats_step() if want_stats else None
    dt, per_ms, n_peaks, nf = w.run(steps, 1)       # (round 5: always a warm-up step -- a context's second scan of a resident reference may build its slot list)
    if want_stats and engine.scan_info()["form"] in ("slot-first", "slot-single"):
        stats = w.stats_step()                      # the counts of the form the timed steps took, not of the first scan's
        stats["slot_list_bytes"] = slot_list_streamed_bytes(engine)
    d = {"value": round(pairs * steps / dt / 1e6, 3), "unit": "M paired-reads/s", "ms_per_step": round(dt / steps * 1e3, 2),
         "phase_ms": {"count_A": round(per_ms[0], 2), "scan_B": round(per_ms[1], 2), "vote_C": round(per_ms[2], 2)},
         "scan_B_form": engine.scan_info(), "peak_registry": engine.registry_info(), "vote_form": engine.vote_info(), "raw_peaks": n_peaks, "filtered_peaks": nf, "steps": steps, "pairs": pairs,
         "work_stats": stats}
    if recall:
        d["planted_transfers"] = interval_recall(out_path, planted_breakpoints(n_contigs, contig_len, sample_contigs))
    roof, dom = rooflines(k, e, L, pairs, ref_bases or n_contigs * contig_len, n_contigs, packed, per_ms, d["scan_B_form"], n_peaks,
                          traffic or {}, LIVE if traffic else None, stats, d["vote_form"])
    d["roofline"] = roof[dom]
    d["roofline_other"] = {ph: r for ph, r in roof.items() if ph != dom}
    d["_shape"] = {"per_ms": per_ms, "k": k, "e": e, "ref_bases": ref_bases or n_contigs * contig_len, "n_contigs": n_contigs, "packed": packed, "L": L}
    return d


def reroof(d, traffic):
    """a leg's rooflines again once its PMC traffic is known (the children run after the engine has let go of the GPU)"""
    s = d["_shape"]
    roof, dom = rooflines(s["k"], s["e"], s.get("L", 150), d["pairs"], s["ref_bases"], s["n_contigs"], s["packed"], s["per_ms"], d["scan_B_form"], d["raw_peaks"],
                          traffic or {}, LIVE if traffic else None, d.get("work_stats"), d.get("vote_form"))
    d["roofline"] = roof[dom]
    d["roofline_other"] = {ph: r for ph, r in roof.items() if ph != dom}


def e2e_from_files(k, e, device, n_contigs=100, contig_len=1_000_000, n_pairs=4_000_000, big_pairs=32_000_000, full=False):
    """from FASTQ files in the page cache through the drop-in entry point (localhgt_amd.extract_ref.run: what bin/extract_ref
    calls): line count (+ sampling ratio), index (built in the first run, loaded in the second), parse + H2D + pack with phase A
    behind it, phases B-D, interval file.  -t 10 as `localhgt bkp` passes it: the reference's thread chunks are emulated.
    Two sizes: 4 M pairs (2.5 GB of text: the fixed costs show) and `big_pairs` (20 GB: the reads decide)."""
    from localhgt_amd import extract_ref
    quiet = dict(device=device, log=lambda *x: None)
    with tempfile.TemporaryDirectory(prefix="lhgt_e2e_") as tmp:
        with near_gpu(device):
            fa, f1, f2 = synth_files(tmp, k, e, n_contigs, contig_len, n_pairs, device)
        a = extract_ref.Args(f1, f2, fa, os.path.join(tmp, "interval.txt"), 0.1, 0.08, 10, k, 300_000_000, e, 1, 1.0)
        reps = [extract_ref.run(a, **quiet) for _ in range(3)]
        built, cached = reps[0], min(reps[1:], key=lambda r: r["total_s"])
        packed = min((extract_ref.run(a, ref_form="packed", **quiet) for _ in range(2)), key=lambda r: r["total_s"])
        fq_bytes = os.path.getsize(f1) + os.path.getsize(f2)
        out = {"value": round(n_pairs / cached["total_s"] / 1e6, 3), "unit": "M paired-reads/s",
               "what": f"extract_ref -t 10 (thread emulation, the CLI default) on {n_pairs} pairs ({fq_bytes / 1e9:.2f} GB of FASTQ, page cache) vs {n_contigs} x {contig_len} bp, "
                       f"k={k} e={e}, cached index; whole call incl. context set-up, index load, parse, H2D, packing, A-D, interval file",
               "total_s": round(cached["total_s"], 3), "ingest_s": round(cached["ingest_s"], 3), "emulated_threads": cached["emulated_threads"],
               "index_load_s": round(cached.get("index_s", 0.0), 3), "reads_s": round(cached.get("reads_s", 0.0), 3),
               "kernels_ms": round(cached["count_kernel_ms"] + cached["scan_kernel_ms"] + cached["vote_kernel_ms"], 1),
               "fastq_GB_per_s": round(fq_bytes / cached["total_s"] / 1e9, 2),
               "with_index_build": {"value": round(n_pairs / built["total_s"] / 1e6, 3), "total_s": round(built["total_s"], 3)},
               "with_packed_reference": {"value": round(n_pairs / packed["total_s"] / 1e6, 3), "total_s": round(packed["total_s"], 3),
                                         "reference_load_s": round(packed.get("index_s", 0.0), 3), "same_peaks": (packed["n_peaks"], packed["n_filtered"]) == (cached["n_peaks"], cached["n_filtered"]),
                                         "what": "LHGT_REF_FORM=packed: no index file read; the FASTA text goes to the GPU, is stripped and packed there, phase B recomputes the hashes"},
               "raw_peaks": cached["n_peaks"], "filtered_peaks": cached["n_filtered"]}
        if full:
            plain = min((extract_ref.run(a, emulate_threads=False, **quiet) for _ in range(2)), key=lambda r: r["total_s"])
            out["without_thread_emulation"] = {"value": round(n_pairs / plain["total_s"] / 1e6, 3), "total_s": round(plain["total_s"], 3),
                                               "what": "LHGT_EMULATE_THREADS=0: the -t 1 result whatever -t says"}
    if big_pairs and shutil.disk_usage(tempfile.gettempdir()).free > 2.2 * 320 * 2 * big_pairs:
        with tempfile.TemporaryDirectory(prefix="lhgt_e2e_") as tmp:
            t0 = time.time()
            with near_gpu(device):                          # the files' pages on the GPU's socket (tools/benchlib/files.py)
                fa, f1, f2 = synth_files_sliced(tmp, k, e, n_contigs, contig_len, big_pairs, device)
            gen_s = time.time() - t0
            fq_bytes = os.path.getsize(f1) + os.path.getsize(f2)
            legs = {}
            todo = [("sample_1", 1.0, {}), ("default_sample_2e9", 2e9, {})]
            if full:
                todo += [("sample_1_packed_reference", 1.0, {"ref_form": "packed"}), ("default_sample_2e9_packed_reference", 2e9, {"ref_form": "packed"})]
            for tag, sample, kw in todo:
                a = extract_ref.Args(f1, f2, fa, os.path.join(tmp, "interval.txt"), 0.1, 0.08, 10, k, 300_000_000, e, 1, sample)
                r = min((extract_ref.run(a, **dict(quiet, **kw)) for _ in range(3 if not legs else 2)), key=lambda r: r["total_s"])
                legs[tag] = {"value": round(big_pairs / r["total_s"] / 1e6, 2), "unit": "M input pairs/s", "total_s": round(r["total_s"], 3),
                             "reads_s": round(r["reads_s"], 3), "reference_s": round(r["index_s"], 3), "pairs_kept": r["pairs_kept"],
                             "ratio_percent": round(r["ratio"], 4), "raw_peaks": r["n_peaks"], "filtered_peaks": r["n_filtered"],
                             "fastq_GB_per_s": round(fq_bytes / r["total_s"] / 1e9, 1)}
            out["big"] = dict(legs, what=f"the same call on {big_pairs} pairs ({fq_bytes / 1e9:.1f} GB of FASTQ in the page cache, written in {gen_s:.0f} s from the CPUs of the GPU's NUMA node), -t 10; "
                                         "default_sample_2e9 = the CLI's default --sample 2000000000 (cal_sam_ratio's base count from the line plan, "
                                         "pairs kept by the sampling array)")
    return out


def e2e_batch(k, e, device, n_contigs=13000, contig_len=1_000_000, n_samples=5, pairs_per_sample=8_000_000, sample_contigs=300, time_budget_s=240):
    """`extract_ref --batch` at the headline's reference size (round 6): n_samples samples as FASTQ files against the 13 Gbase
    reference as a FASTA file, through localhgt_amd.extract_ref.run_batch -- what `bin/extract_ref --batch MANIFEST` calls -- with
    LHGT_REF_FORM=packed: ONE context, the reference loaded by the first sample, its slot list built before the second sparse scan (the
    engine's own rule), both kept for the rest.  Every sample's 12 arguments are those of scripts/pipeline.sh:35 (-t 10, sample = 1).
    Reported: input pairs/s over the whole batch INCLUDING the reference load and the list build, the per-sample times, and the same
    first sample as a call of its own (what one process per sample pays every time)."""
    from localhgt_amd import extract_ref
    from localhgt_amd.engine import Engine
    from .files import write_fasta, write_fastq
    t_all = time.time()
    need = 1.1 * (n_contigs * (contig_len + 8) + n_samples * 2 * 320 * pairs_per_sample)
    if shutil.disk_usage(tempfile.gettempdir()).free < need:
        return {"skipped": f"needs {need / 1e9:.0f} GB of scratch space"}
    with tempfile.TemporaryDirectory(prefix="lhgt_batch_") as tmp:
        fa = os.path.join(tmp, "ref.fa")
        samples = []
        with near_gpu(device), Engine(k, e, device=device) as eng:
            eng.rng_seed(1)
            eng.coder_generate()
            eng.set_reference_form(True)
            write_fasta(fa, eng.synth_reference(1, n_contigs, contig_len, want_host=True), n_contigs, contig_len)
            eng.synth_options(0, 20, sample_contigs)
            for i in range(n_samples):
                eng.pairs_clear()
                m1, m2 = eng.synth_pairs(1, 100 + i, n_contigs, contig_len, 0, pairs_per_sample, 150, want_host=True)
                f1, f2 = os.path.join(tmp, f"s{i}.1.fq"), os.path.join(tmp, f"s{i}.2.fq")
                write_fastq(f1, m1, pairs_per_sample, 150, "1")
                write_fastq(f2, m2, pairs_per_sample, 150, "2")
                samples.append(extract_ref.Args(f1, f2, fa, os.path.join(tmp, f"s{i}.interval.txt"), 0.1, 0.08, 10, k, 300_000_000, e, 1, 1.0))
        gen_s = time.time() - t_all
        fq_bytes = sum(os.path.getsize(a.fq1) + os.path.getsize(a.fq2) for a in samples)
        quiet = lambda *x: None                                   # noqa: E731
        t0 = time.time()
        reps = extract_ref.run_batch(samples, device=device, log=quiet, ref_form="packed")
        batch_s = time.time() - t0
        outs = [open(a.interval, "rb").read() for a in samples]
        single = None
        if time.time() - t_all < time_budget_s:                   # the first sample again as a process-per-sample call pays it (warm page cache, like the batch's later samples)
            os.remove(samples[0].interval)
            single = extract_ref.run(samples[0], device=device, log=quiet, ref_form="packed")
            single["same_file"] = open(samples[0].interval, "rb").read() == outs[0]
        per = [{"total_s": round(r["total_s"], 3), "reference_s": round(r["index_s"], 3), "reads_s": round(r["reads_s"], 3), "scan_s": round(r["scan_s"], 3),
                "vote_s": round(r["vote_s"], 3), "scan_form": r["scan_form"], "ref_reused": r["ref_reused"], "raw_peaks": r["n_peaks"], "filtered_peaks": r["n_filtered"],
                "slot_list_GB": round(r["slot_list_bytes"] / 1e9, 1)} for r in reps]
        steady = [p["total_s"] for p in per[2:]] or [per[-1]["total_s"]]
        out = {"value": round(n_samples * pairs_per_sample / batch_s / 1e6, 2), "unit": "M input pairs/s",
               "what": f"extract_ref --batch: {n_samples} samples x {pairs_per_sample} pairs (FASTQ files, {fq_bytes / 1e9:.1f} GB, page cache; each drawn from {sample_contigs} genomes, sample = 1, -t 10) "
                       f"against the {n_contigs * contig_len / 1e9:.0f} Gbase reference as a FASTA file, LHGT_REF_FORM=packed, ONE process and context; the time INCLUDES the "
                       "reference load (first sample) and the slot list build (second sample)",
               "batch_s": round(batch_s, 3), "samples": per, "steady_sample_s": round(sum(steady) / len(steady), 3),
               "steady_input_pairs_per_s_M": round(pairs_per_sample / (sum(steady) / len(steady)) / 1e6, 2),
               "files_written_s": round(gen_s, 1)}
        if single:
            out["one_process_per_sample"] = {"total_s": round(single["total_s"], 3), "reference_s": round(single["index_s"], 3), "scan_s": round(single["scan_s"], 3),
                                             "scan_form": single["scan_form"], "value": round(pairs_per_sample / single["total_s"] / 1e6, 2),
                                             "same_interval_file_as_in_the_batch": single["same_file"]}
            gain = single["total_s"] - out["steady_sample_s"]
            extra = sum(p["total_s"] for p in per[:2]) - 2 * single["total_s"]       # what the first two samples of the batch cost beyond two calls of their own
            out["break_even_samples"] = round(2 + max(0.0, extra) / gain, 1) if gain > 0 else None
    return out


def pipelined_samples(k, e, device, n_contigs, contig_len, pairs, n_samples=4):
    """(--full only; measured in round 3: gain 0.998.)  Two contexts on one GPU (each with its own stream, tables and read store;
    the reference resident as packed bases in both), two host threads: a sample's phase A may run while the other context is in
    its phases B-D, never two of the same kind at once.  Reported: pairs/s over n_samples samples against the same samples one
    after the other, per-phase kernel times in both modes, and whether every sample's peaks and vote table are the serial run's."""
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
