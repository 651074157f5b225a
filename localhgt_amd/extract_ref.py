"""`extract_ref` -- drop-in for the reference binary at scripts/pipeline.sh:35.

    extract_ref fq1 fq2 ref.fa interval_out hit_ratio match_ratio threads k max_peak e seed sample

Same 12 positional arguments, parsed the way the reference does (stod then truncation,
/root/reference/src/extract_ref_normal_peak.cpp:1352-1371), same files read and written:
`<ref>.k<k>.h<e>.index.dat` + `<ref>.genome.len.txt` (built when absent, reused when present,
E:1401-1413) and the interval file (E:515-548).  All compute runs on the GPU through
liblocalhgt_hip.so; `threads` does not set any thread count here, but with threads > 1 the RESULT is the one of the
reference's `-t threads` run (its read partition, id ranges and sentinel lines, restated without its races;
LHGT_EMULATE_THREADS=0: the `-t 1` result whatever -t says).

Under `torch.distributed.run` (WORLD_SIZE > 1) every rank takes a contiguous run of the read pairs and
the count table / votes are exchanged over RCCL -- or, when the ranks share a GPU, through host memory over gloo
(localhgt_amd/dist.py)."""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass

import numpy as np

from .engine import Engine


@dataclass
class Args:
    fq1: str
    fq2: str
    fasta: str
    interval: str
    hit_ratio: float
    match_ratio: float
    threads: int
    k: int
    max_peak: int
    e: int
    seed: int
    sample: float


def parse_argv(argv) -> Args:
    if len(argv) < 12:
        raise SystemExit("usage: extract_ref fq1 fq2 ref.fa interval_out hit_ratio match_ratio threads k max_peak "
                         "coder_num seed base_num")
    f = [float(x) for x in argv[4:12]]  # stod
    return Args(argv[0], argv[1], argv[2], argv[3], float(np.float32(f[0])), float(np.float32(f[1])), int(f[2]),
                int(f[3]), int(f[4]), int(f[5]), int(f[6]) & 0xFFFFFFFF, f[7])


def index_name(fasta: str, k: int, e: int) -> str:
    return f"{fasta}.k{k}.h{e}.index.dat"  # E:1401


def _warn_if_ids_desynchronise(genome_len_path: str, k: int, log) -> None:
    """SURVEY.md quirk Q7, told BEFORE any GPU work: read_ref numbers every FASTA record (E:825) while read_index numbers the
    indexed contigs sequentially (E:905, 963), so a contig of <= k bases in front of an indexed one makes the ids of the interval
    file point at the wrong line of genome.len.txt.  The interval file is still the reference's; get_bed_file will refuse it."""
    try:
        with open(genome_len_path) as f:
            ids = [int(line.split("\t")[1]) for line in f if line.strip()]
    except (OSError, ValueError, IndexError):
        return
    if ids != list(range(ids[0] if ids else 0, (ids[0] if ids else 0) + len(ids))) or (ids and ids[0] != 1):
        log(f"warning: {genome_len_path} skips contig numbers (a contig of <= {k} bases precedes an indexed one): the contig ids of "
            f"the interval file follow the reference's sequential numbering (E:905) and will not match; get_bed_file refuses them")


def _emulation_refused(err) -> bool:
    """the error with which only the -t N emulation refuses an input (include/localhgt_hip.h: LHGT_E_EMULATION)"""
    return err.code == 9


class Session:
    """One process that handles sample after sample (round 6: `extract_ref --batch`).  The reference runs one sample per process
    (scripts/pipeline.sh:35) and pays the reference load every time; a session keeps ONE context per (k, e) and with it whatever
    a next sample of the same reference can reuse: the resident reference (index hashes or packed bases), the slot list the
    second sparse-table scan of that reference builds (include/localhgt_hip.h: lhgt_slot_list), the 16 GiB peak_kmer table and
    the key buffers.  Every sample still gets exactly the files a process of its own would have written -- the per-sample state
    (seed, coder draws, sampling array, reads, count table, thread emulation) is set up from scratch as `run` does."""

    def __init__(self, device: int = 0, dist=None, log=print):
        self.device, self.dist, self.log = device, dist, log
        self.eng = None
        self.resident = None        # identity of the reference the context holds: (path, size, mtime_ns, form, shard) [+ the index file's]
        self.ref_shape = (0, 0)     # its (contigs, bases)
        self.samples = 0

    def _engine(self, k: int, e: int) -> Engine:
        if self.eng is not None and (self.eng.k, self.eng.e) != (k, e):
            self.close()
        if self.eng is None:
            self.eng = Engine(k, e, self.device)
            self.resident = None
        return self.eng

    def close(self):
        if self.eng is not None:
            self.eng.close()
        self.eng, self.resident = None, None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def run(self, a: Args, emulate_threads=None, ref_form=None, log=None) -> dict:
        t0 = time.time()
        eng = self._engine(a.k, a.e)
        try:
            rep = _run(eng, a, t0, self.dist, log or self.log, emulate_threads, ref_form, self)
        except BaseException:
            self.close()     # also on an error: a long-lived caller must not keep a 16 GiB peak_kmer table (and the reads) per failed
            raise            # call, and the next sample must not inherit half a sample's state
        self.samples += 1
        return rep


def run(a: Args, device: int = 0, dist=None, log=print, emulate_threads=None, ref_form=None) -> dict:
    """The whole path A->D. `dist` is a localhgt_amd.dist.Exchange (or None for one GPU).
    ref_form (default: LHGT_REF_FORM in the environment, "index"): "packed" keeps the reference's BASES resident (3/8 byte per
    base, read from the FASTA) instead of the index file's hashes (12 bytes per base at e = 3) and lets phase B recompute the
    hashes: same interval file, no 12-bytes-per-base file read or written -- only the coder header of an existing index is
    used; without one the coder is drawn as the index build would draw it and genome.len.txt is written.
    emulate_threads (default: on when threads > 1, LHGT_EMULATE_THREADS=0 in the environment turns it off): give the result of
    the reference's `-t threads` run without its races -- its per-thread read partition, id ranges and sentinel lines (SURVEY.md
    8f rank 4) -- instead of the `-t 1` result.  `localhgt bkp` passes -t 10 by default, so that is what a user's reference run
    produced (under sampling the kept reads, and with them the .bed, differ between -t 1 and -t N, E:1037).  Where the emulation
    refuses an input -- one on which the reference reads stale bytes or overruns a thread's id range -- the run falls back to
    the -t 1 result with one warning line."""
    with Session(device, dist, log) as ses:
        return ses.run(a, emulate_threads, ref_form)


def _file_id(path: str):
    st = os.stat(path)
    return (os.path.realpath(path), st.st_size, st.st_mtime_ns)


def _run(eng: Engine, a: Args, t0: float, dist, log, emulate_threads, ref_form, ses: Session) -> dict:
    from ._lib import LocalHGTError
    rank, world = (dist.rank, dist.world) if dist else (0, 1)
    if ses.samples:                                            # a context that has seen a sample: nothing of it may reach this one
        eng.pairs_clear()
        eng.counts_clear()
        eng.sampling_reserve(0)                                # (the entries the LAST sample's reads could look at: get_random fills all of them until told otherwise)
    if emulate_threads is None:
        emulate_threads = os.environ.get("LHGT_EMULATE_THREADS", "1") != "0"
    emulating = bool(emulate_threads) and a.threads > 1
    eng.set_thread_emulation(a.threads if emulating else 1)
    if emulating:
        log(f"reproducing the reference's -t {a.threads} read partition and peak id ranges")
    log(f"kmer length is {a.k}\nseed is {a.seed}\nnum of hash functions is {a.e}")
    eng.rng_seed(a.seed)                                       # E:1386
    idx = index_name(a.fasta, a.k, a.e)
    if ref_form is None:
        ref_form = os.environ.get("LHGT_REF_FORM", "index")
    if ref_form not in ("index", "packed"):
        raise SystemExit(f"LHGT_REF_FORM: 'index' or 'packed', not {ref_form!r}")
    packed = ref_form == "packed"
    # E:1403-1410.  random_coder draws k*(e//3+1) values from the rand() stream before the sampling
    # array is filled (quirk Q3), so every rank draws them when rank 0 has to build the index.
    built = not os.path.exists(idx) if rank == 0 else False
    if dist:
        built = dist.broadcast_flag(built)
    if built:
        eng.coder_generate()
    # get_random's 5*10^7 draws (E:1422, 0.3 s of one host core) depend on nothing but the seed and the coder's draws: they start
    # here, on a thread of their own, next to the line count and the reference load, and are joined by sampling_init below --
    # which cuts them short once the number of reads is known, or drops them when the ratio turns out >= 100 % (nothing looks).
    # With --sample 1 exactly the ratio is known to be 100 % and nothing is drawn.
    if a.sample != 1 and os.environ.get("LHGT_SYNC_SAMPLING", "0") != "1":   # =1: draw them where the reference does (A/B timing)
        eng.sampling_begin()
    plan1 = plan2 = None
    from . import pack as _pack
    packed_sample = _pack.read_header(a.fq1) if _pack.is_packed(a.fq1) else None    # a sample packed by localhgt_pack (fq2 is then ignored: "-")
    if packed_sample:
        log(f"reads: {a.fq1} is a packed sample ({packed_sample.n_pairs} pairs, {packed_sample.stride} bytes each)")
    elif dist:                                                 # every rank counts the lines of 1/world of both files (dist.fastq_plan)
        ch1, ch2 = eng.fastq_pair_chunks(a.fq1, a.fq2)          # fq2 in as many chunks as fq1: the parse can then take whole columns
        plan1 = dist.fastq_plan(eng, a.fq1, want_len_sums=a.sample > 1, chunk=ch1)
        plan2 = dist.fastq_plan(eng, a.fq2, chunk=ch2)
    elif a.sample > 1:                                         # the CLI's default --sample 2000000000: the line count the loader needs
        plan1, plan2 = eng.fastq_plan(a.fq1, True, other=a.fq2)   # anyway also yields cal_sam_ratio's base count (no extra pass over fq1)
    if packed_sample:
        ratio = packed_sample.ratio(a.sample)
    else:
        ratio = eng.sam_ratio_from_plan(plan1, a.sample) if plan1 is not None else eng.sam_ratio(a.fq1, a.sample)   # E:1392-1398
    log(f"down-sampling ratio: {ratio}%.")
    if built:
        if rank == 0 and not packed:
            log("Reference index not detected, start index...")
            eng.index_build(a.fasta, idx, a.fasta + ".genome.len.txt")
        if dist and not packed:
            dist.barrier()
    elif packed:
        eng.index_read_coder(idx)                              # E:1413: the coder the index was built with
    # Phase B form: replicated index (default) or reference-sharded (LHGT_SHARD_INDEX=1, or =auto when the index
    # does not fit next to the other tables of one GPU): see localhgt_amd/dist.py
    shard_index = False
    if dist and not packed:
        mode = os.environ.get("LHGT_SHARD_INDEX", "auto")
        shard_index = mode == "1" or (mode == "auto" and os.path.getsize(idx) > 180e9)
    # (Uploading the index on a second host thread next to the FASTQ pipeline was tried and lost: 0.23 s instead of 0.17 s for
    # 4 M pairs + a 1.2 GB index -- the page faults and the pinning of the index mapping fight the parse threads.)
    t_i0 = time.time()
    # a session's next sample of the same reference finds it resident (same file by path, size and mtime; same form and shard)
    want = _file_id(a.fasta) + (ref_form, (rank, world) if shard_index else None) + (() if packed else _file_id(idx))
    reused = ses.resident == want
    if reused:
        n_contigs, n_bases = ses.ref_shape
        log("reference: resident from the previous sample of this session")
        if packed and built and rank == 0 and not os.path.exists(a.fasta + ".genome.len.txt"):
            eng.fasta_scan(a.fasta, a.fasta + ".genome.len.txt")   # what a run of its own would have (re)written (E:773, 878)
    else:
        ses.resident = None
        if packed:
            eng.set_reference_form(True)
            log("reference form: packed bases from the FASTA, hashes recomputed in the scan")
            n_contigs, n_bases = eng.reference_load_fasta(a.fasta, a.fasta + ".genome.len.txt" if built and rank == 0 else None)
        elif shard_index:
            eng.set_reference_form(False)
            n_contigs, n_bases = eng.index_load_shard(idx, rank, world)
        else:
            eng.set_reference_form(False)
            n_contigs, n_bases = eng.index_load(idx)           # E:1417 (+ resident copy of the hashes)
        ses.resident, ses.ref_shape = want, (n_contigs, n_bases)
    if packed and dist and built:
        dist.barrier()
    t_i1 = time.time()
    _warn_if_ids_desynchronise(a.fasta + ".genome.len.txt", a.k, log)
    if packed_sample:
        eng.sampling_reserve(packed_sample.n_pairs)
    elif plan1 is not None:                                    # reads per file are known: only the entries they can look at are filled
        eng.sampling_reserve(max((int(plan1[1].sum()) + 2) // 4, (int(plan2[1].sum()) + 2) // 4))
    eng.sampling_init(ratio)                                   # E:1422
    t_r0 = time.time()
    eng.set_count_on_load(True)                                # phase A of a batch runs while the next one is parsed (E:1426-1448)
    state = {"seen": 0, "kept": 0, "t_reads": 0.0}

    def everywhere(fn):
        """fn() on this rank -- rank-local work only, no collective inside -- and its outcome agreed over the ranks before anyone
        goes on: 0 fine, 1 the -t N emulation refuses the input (returned: the caller falls back on every rank), 2 any other
        error (raised on EVERY rank: one that failed alone -- an over-long read in its byte range, a HIP error -- would otherwise
        leave the others waiting in the next collective until the backend's timeout)"""
        err, mine = None, 0
        try:
            fn()
        except LocalHGTError as e:
            err, mine = e, (1 if emulating and _emulation_refused(e) else 2)
        except Exception as e:                                 # noqa: BLE001 -- agreed on first, re-raised below
            err, mine = e, 2
        worst = dist.agree(mine) if dist else mine
        if worst == 2:
            if mine == 2:
                raise err
            raise LocalHGTError(5, f"another rank failed (rank {rank} stops with it)")
        if worst == 1:
            return err if mine == 1 else LocalHGTError(9, "-t N emulation: refused on another rank")
        return None

    def load_and_count():
        t = time.time()
        if packed_sample:                                      # this rank's contiguous run of the packed pairs; the GPU decides which the run keeps
            _, state["kept"] = eng.pairs_load_packed(packed_sample, ratio, a.threads if emulating else 1, rank, world)
            state["seen"] = packed_sample.n_pairs
        elif plan1 is not None:                                # this rank's contiguous run of fq1's chunks, paired through the whole plan
            state["seen"], state["kept"] = eng.pairs_load_fastq_planned(a.fq1, a.fq2, ratio, plan1[:2], plan2[:2], rank, world)
        else:
            state["seen"], state["kept"] = eng.pairs_load_fastq(a.fq1, a.fq2, ratio)
        state["t_reads"] = time.time() - t
        eng.count_kmers()                                      # phase A, E:1426-1448

    def merge_counts():                                        # the first collective of the run: entered only once every rank has loaded
        if dist:
            dist.merge_counts(eng)

    def fall_back(err):
        nonlocal emulating
        log(f"warning: the reference's -t {a.threads} run is not defined on this input ({str(err).split(': ', 2)[-1]}); "
            f"giving the -t 1 result")
        emulating = False
        eng.set_thread_emulation(1)
        eng.pairs_clear()
        eng.counts_clear()

    err = everywhere(load_and_count)
    if err:
        fall_back(err)
        everywhere(load_and_count)
    merge_counts()
    t2 = time.time()
    log(f"K-mer counting is finished. It costs {t2 - t0:.2f} seconds.")
    scan = {}

    def scan_ref():
        if shard_index:                                        # phase B, E:1468-1489
            scan["n"] = dist.sharded_scan(eng, a.hit_ratio, a.match_ratio, a.max_peak, a.threads if emulating else 1)
        else:
            scan["n"] = eng.ref_scan(a.hit_ratio, a.match_ratio, a.max_peak)

    # (the sharded scan holds collectives itself and agrees inside dist.sharded_scan before its first exchange; the refusals of the
    # -t N emulation there come from global sums, the same on every rank)
    err = _collective(scan_ref, emulating, LocalHGTError) if shard_index else everywhere(scan_ref)
    if err:                                                    # a thread's peaks overflow its id range: reads again, as -t 1 keeps them
        fall_back(err)
        everywhere(load_and_count)
        merge_counts()
        scan_ref()
    n_peaks = scan["n"]
    seen, kept = state["seen"], state["kept"]
    t3 = time.time()
    log(f"Slided ref len: {n_bases} bp\tNo. of raw BKPs: {n_peaks}")
    eng.vote()                                                 # phase C, E:1496-1507
    if dist:
        dist.sum_votes(eng)
    t4 = time.time()
    n_filtered = -1
    if rank == 0:
        n_filtered = eng.write_intervals(a.interval)           # phase D, E:1511
    if dist:
        dist.barrier()
    t5 = time.time()
    log(f"Finish with time:\t{t5 - t0:.2f}")
    rep = dict(pairs_seen=seen, pairs_kept=kept, n_contigs=n_contigs, n_bases=n_bases, n_peaks=n_peaks,
               n_filtered=n_filtered, ratio=ratio, index_built=built and not packed, ref_form=ref_form, emulated_threads=a.threads if emulating else 1,
               ref_resident_bytes=eng.reference_info()["resident_bytes"], ref_reused=reused, scan_form=eng.scan_info()["form"], registry_chunks=eng.registry_info()["chunks"], slot_list_bytes=eng.slot_list()["bytes"], ingest_s=t_r0 + state["t_reads"] - t0, index_s=t_i1 - t_i0,
               reads_s=state["t_reads"], count_s=t2 - t_r0 - state["t_reads"], scan_s=t3 - t2,
               vote_s=t4 - t3, total_s=t5 - t0, count_kernel_ms=eng.phase_ms(0), scan_kernel_ms=eng.phase_ms(1),
               vote_kernel_ms=eng.phase_ms(2), world=world, staged_bytes=dist.staged_bytes if dist else 0)
    return rep


def _collective(fn, emulating, LocalHGTError):
    """fn() holds collectives itself (the reference-sharded scan): the refusal of the -t N emulation is returned, anything else
    propagates"""
    try:
        fn()
    except LocalHGTError as e:
        if not (emulating and _emulation_refused(e)):
            raise
        return e
    return None


def read_manifest(path: str):
    """one sample per line: the 12 arguments of an `extract_ref` call as scripts/pipeline.sh:35 passes them (shell quoting allowed,
    `#` starts a comment, blank lines are skipped)"""
    import shlex
    samples = []
    with open(path) as f:
        for no, line in enumerate(f, 1):
            tok = shlex.split(line, comments=True)
            if not tok:
                continue
            if len(tok) != 12:
                raise SystemExit(f"{path}:{no}: {len(tok)} arguments, an extract_ref call has 12 "
                                 "(fq1 fq2 ref.fa interval_out hit_ratio match_ratio threads k max_peak coder_num seed base_num)")
            samples.append(parse_argv(tok))
    return samples


def run_batch(samples, device: int = 0, dist=None, log=print, ref_form=None, emulate_threads=None, keep_going=True):
    """the samples one after the other in ONE session (class Session): the files of every sample are those of a call of its
    own.  A sample that fails is reported and -- on one GPU -- the batch goes on with a fresh context; returns the per-sample
    reports (None for a failed one)."""
    reps = []
    with Session(device, dist, log) as ses:
        for i, a in enumerate(samples):
            log(f"---- batch sample {i + 1} of {len(samples)}: {a.interval}")
            try:
                reps.append(ses.run(a, emulate_threads, ref_form))
            except Exception as ex:                            # noqa: BLE001 -- reported; the others still run
                if dist or not keep_going:
                    raise
                log(f"error: sample {i + 1} ({a.interval}) failed: {ex}")
                reps.append(None)
    return reps


def main(argv=None) -> int:
    """extract_ref <12 arguments>             one sample, as scripts/pipeline.sh:35 calls it
    extract_ref --batch MANIFEST            one process for many samples: MANIFEST holds the 12 arguments of each call, one per line;
                                            the reference of consecutive samples stays resident on the GPU (round 6)"""
    argv = sys.argv[1:] if argv is None else argv
    batch = None
    if argv and argv[0] == "--batch":
        if len(argv) != 2:
            raise SystemExit("usage: extract_ref --batch MANIFEST   (one extract_ref argument list per line)")
        batch = read_manifest(argv[1])
    else:
        a = parse_argv(argv)
    dist = None
    device = 0
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from .dist import Exchange
        dist = Exchange.from_env()
        device = dist.device
    try:
        if batch is not None:
            reps = run_batch(batch, device=device, dist=dist)
            return 1 if any(r is None for r in reps) else 0
        run(a, device=device, dist=dist)
    finally:
        if dist:
            dist.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
