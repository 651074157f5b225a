"""Python face of the C-ABI: one `Engine` = one `lhgt_ctx` on one GPU.

Method names follow the phases of the reference binary
(/root/reference/src/extract_ref_normal_peak.cpp: main at :1342-1519)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def pool_trim() -> None:
    """free the tables that closed contexts left parked for reuse (include/localhgt_hip.h: lhgt_pool_trim)"""
    _lib.check(_lib.load(require_gpu=False).lhgt_pool_trim())


class Engine:
    def __init__(self, k: int, e: int, device: int = 0):
        self.lib = _lib.load(require_gpu=(device >= 0))  # device -1 = host-only context (RNG/coder rows)
        self.k, self.e, self.device = int(k), int(e), int(device)
        h = C.c_void_p()
        _lib.check(self.lib.lhgt_ctx_create(self.device, self.k, self.e, C.byref(h)))
        self.h = h
        self.emulated_threads = 1

    def close(self):
        """frees the context.  One thing outlives it: a peak_kmer table of 4 GiB or more (16 GiB at k = 32) is parked for the next
        context of this process on the same device (hipMalloc of 16 GiB right after such a free took up to 2 s); it is handed back
        by `localhgt_amd.engine.pool_trim()`, and by itself whenever a device allocation would otherwise run out of memory"""
        if getattr(self, "h", None):
            self.lib.lhgt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- R
    def rng_seed(self, seed: int):
        _lib.check(self.lib.lhgt_rng_seed(self.h, C.c_uint(seed & 0xFFFFFFFF)))

    def coder_generate(self):
        _lib.check(self.lib.lhgt_coder_generate(self.h))

    def coder_set(self, cc: np.ndarray):
        cc = np.ascontiguousarray(cc, dtype=np.int16)
        assert cc.size == _lib.CODER_SLOTS
        _lib.check(self.lib.lhgt_coder_set(self.h, _ptr(cc, C.c_int16)))

    def coder_get(self) -> np.ndarray:
        cc = np.zeros(_lib.CODER_SLOTS, dtype=np.int16)
        _lib.check(self.lib.lhgt_coder_get(self.h, _ptr(cc, C.c_int16)))
        return cc

    def sampling_init(self, ratio_percent: float):
        _lib.check(self.lib.lhgt_sampling_init(self.h, float(ratio_percent)))

    def sampling_begin(self):
        """start filling the sampling array on a host thread (after the coder's draws); sampling_init(ratio) joins it"""
        _lib.check(self.lib.lhgt_sampling_begin(self.h))

    def sampling_reserve(self, n_reads: int):
        """the run will see at most n_reads reads per file: sampling_init fills only that many entries of the sampling array"""
        _lib.check(self.lib.lhgt_sampling_reserve(self.h, int(n_reads)))

    def sampling_get(self, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.float32)
        _lib.check(self.lib.lhgt_sampling_get(self.h, _ptr(out, C.c_float), n))
        return out

    # ---- H
    def hash_sequence(self, seq: bytes) -> Tuple[np.ndarray, np.ndarray]:
        nk = max(0, len(seq) - self.k + 1)
        out = np.zeros((nk, self.e), dtype=np.uint32)
        valid = np.zeros(nk, dtype=np.uint8)
        _lib.check(self.lib.lhgt_hash_sequence(self.h, seq, len(seq), _ptr(out, C.c_uint32), _ptr(valid, C.c_uint8)))
        return out, valid.astype(bool)

    # ---- I
    def index_build(self, fasta: str, index_path: str, genome_len_path: str) -> Tuple[int, int]:
        nc, nb = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_index_build(self.h, fasta.encode(), index_path.encode(), genome_len_path.encode(),
                                             C.byref(nc), C.byref(nb)))
        return nc.value, nb.value

    def index_load(self, index_path: str) -> Tuple[int, int]:
        nc, nb = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_index_load(self.h, index_path.encode(), C.byref(nc), C.byref(nb)))
        return nc.value, nb.value

    def index_load_shard(self, index_path: str, rank: int, world: int) -> Tuple[int, int]:
        nc, nb = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_index_load_shard(self.h, index_path.encode(), rank, world, C.byref(nc), C.byref(nb)))
        return nc.value, nb.value

    def set_reference_form(self, packed: bool):
        """False: the index file's hashes resident (4e bytes per base).  True: the bases resident as bit-planes (3/8 byte per
        base), phase B recomputes the hashes.  A resident reference of the other form is dropped."""
        _lib.check(self.lib.lhgt_set_reference_form(self.h, 1 if packed else 0))

    def reference_info(self) -> dict:
        form, nbytes = C.c_int(0), C.c_uint64(0)
        _lib.check(self.lib.lhgt_reference_info(self.h, C.byref(form), C.byref(nbytes)))
        return {"form": "packed" if form.value else "index", "resident_bytes": nbytes.value}

    def index_read_coder(self, index_path: str):
        _lib.check(self.lib.lhgt_index_read_coder(self.h, index_path.encode()))

    def reference_load_fasta(self, fasta: str, genome_len_path: Optional[str] = None) -> Tuple[int, int]:
        nc, nb = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_reference_load_fasta(self.h, fasta.encode(), genome_len_path.encode() if genome_len_path else None,
                                                      C.byref(nc), C.byref(nb)))
        return nc.value, nb.value

    def fasta_scan(self, fasta: str, genome_len_path: Optional[str] = None) -> Tuple[int, int, int]:
        """host-only: (sequences, indexed contigs, their bases) of a FASTA as read_ref sees it; writes genome.len.txt when asked"""
        ns, nc, nb = C.c_long(0), C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_fasta_scan(fasta.encode(), self.k, genome_len_path.encode() if genome_len_path else None,
                                            C.byref(ns), C.byref(nc), C.byref(nb)))
        return ns.value, nc.value, nb.value

    def index_from_memory(self, ascii_bases: np.ndarray, offsets: np.ndarray):
        a = np.ascontiguousarray(ascii_bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        _lib.check(self.lib.lhgt_index_from_memory(self.h, _ptr(a, C.c_uint8), _ptr(o, C.c_uint64), o.size - 1))

    # ---- reads
    def sam_ratio(self, fq1: str, sample: float) -> float:
        r, n = C.c_double(0), C.c_long(0)
        _lib.check(self.lib.lhgt_fastq_sam_ratio(fq1.encode(), float(sample), C.byref(r), C.byref(n)))
        return r.value

    def pairs_load_fastq(self, fq1: str, fq2: str, ratio_percent: float, rank: int = 0, world: int = 1,
                         block: int = 4096) -> Tuple[int, int]:
        seen, kept = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_pairs_load_fastq(self.h, fq1.encode(), fq2.encode(), float(ratio_percent), rank, world,
                                                  block, C.byref(seen), C.byref(kept)))
        return seen.value, kept.value

    # ---- multi-GPU ingest: every rank counts the lines of its share of a file, the pieces are exchanged (localhgt_amd/dist.py)
    def fastq_pair_chunks(self, fq1: str, fq2: str) -> Tuple[int, int]:
        """chunk sizes at which to plan the two files so that the planned parse can take the single-pass loader's columns"""
        c1, c2 = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_fastq_pair_chunk_bytes(fq1.encode(), fq2.encode(), C.byref(c1), C.byref(c2)))
        return c1.value, c2.value

    def fastq_plan_part(self, path: str, part: int, parts: int, want_len_sums: bool = False, chunk: Optional[int] = None):
        """(start[n] u64, n_lines[n] i64, len_sums[n, 4] i64 or None) of this part's chunks (include/localhgt_hip.h)"""
        chunk = chunk or self.lib.lhgt_fastq_plan_chunk_bytes()
        n, tot = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_fastq_plan_part(path.encode(), chunk, part, parts, None, None, 0, C.byref(n), C.byref(tot), None))
        st, cn = np.zeros(max(n.value, 1), dtype=np.uint64), np.zeros(max(n.value, 1), dtype=np.int64)
        sums = np.zeros((max(n.value, 1), 4), dtype=np.int64) if want_len_sums else None
        _lib.check(self.lib.lhgt_fastq_plan_part(path.encode(), chunk, part, parts, _ptr(st, C.c_uint64), _ptr(cn, C.c_long), n.value,
                                                 C.byref(n), C.byref(tot), None if sums is None else _ptr(sums, C.c_long)))
        return st[:n.value], cn[:n.value], None if sums is None else sums[:n.value]

    def fastq_plan(self, path: str, want_len_sums: bool = False, other: Optional[str] = None):
        """the whole plan of one file made here (one rank); with `other`, that file's plan is made at the same time on a second
        thread -- cut into as many chunks as `path` (fastq_pair_chunks) -- and (plan, plan_other) is returned"""
        if other is None:
            return self.fastq_plan_part(path, 0, 1, want_len_sums)
        import threading
        box = {}
        ch1, ch2 = self.fastq_pair_chunks(path, other)
        t = threading.Thread(target=lambda: box.update(p2=self.fastq_plan_part(other, 0, 1, False, chunk=ch2)))
        t.start()
        p1 = self.fastq_plan_part(path, 0, 1, want_len_sums, chunk=ch1)
        t.join()
        return p1, box["p2"]

    @staticmethod
    def sam_ratio_from_plan(plan, sample: float) -> float:
        """cal_sam_ratio (E:1244-1270, 1392-1398) without its pass over fq1: the bases of the sequence lines -- global line index
        % 4 == 1 -- from the per-chunk sums of line lengths by local line index (lhgt_fastq_plan_part: len_sums)"""
        if sample <= 1:
            return 100.0 * sample
        _, counts, sums = plan
        line0 = np.concatenate([[0], np.cumsum(counts)[:-1]]) if len(counts) else np.zeros(0, dtype=np.int64)
        bases = int(sums[np.arange(len(counts)), (1 - line0) % 4].sum()) if len(counts) else 0
        if bases == 0:               # an empty fq1 (or only empty sequence lines): the reference divides by zero in floating point
            return float("inf")      # (E:1266) and keeps every read; so does lhgt_fastq_sam_ratio
        return 100.0 * sample / (2.0 * bases)

    def pairs_load_fastq_planned(self, fq1: str, fq2: str, ratio_percent: float, plan1, plan2, part: int, parts: int) -> Tuple[int, int]:
        """plan = (start, n_lines) of ALL chunks of the file, the parts' pieces in order; only part `part`'s run of fq1's chunks is parsed"""
        seen, kept = C.c_long(0), C.c_long(0)
        s1, c1 = np.ascontiguousarray(plan1[0], dtype=np.uint64), np.ascontiguousarray(plan1[1], dtype=np.int64)
        s2, c2 = np.ascontiguousarray(plan2[0], dtype=np.uint64), np.ascontiguousarray(plan2[1], dtype=np.int64)
        _lib.check(self.lib.lhgt_pairs_load_fastq_planned(self.h, fq1.encode(), fq2.encode(), float(ratio_percent),
                                                          _ptr(s1, C.c_uint64), _ptr(c1, C.c_long), s1.size,
                                                          _ptr(s2, C.c_uint64), _ptr(c2, C.c_long), s2.size, part, parts,
                                                          C.byref(seen), C.byref(kept)))
        return seen.value, kept.value

    def pairs_load_packed(self, hdr, ratio_percent: float, threads: int = 1, part: int = 0, parts: int = 1) -> Tuple[int, int]:
        """part `part` of `parts` of a packed sample (localhgt_amd/pack.py: Header) becomes resident; threads > 1: the reference's -t
        threads read partition from the header's thread chunks (raises LocalHGTError 9 where the emulation refuses the files)"""
        seen, kept = C.c_long(0), C.c_long(0)
        tabs = [None] * 4
        if threads > 1:
            tabs = [np.ascontiguousarray(x, dtype=np.int64) for x in hdr.thread_chunks(threads)]
        ptr = [None if t is None else _ptr(t, C.c_long) for t in tabs]
        _lib.check(self.lib.lhgt_pairs_load_packed(self.h, hdr.path.encode(), hdr.data_offset, hdr.stride, hdr.n_pairs, hdr.q4_first_pair,
                                                   float(ratio_percent), int(threads), ptr[0], ptr[1], ptr[2], ptr[3], part, parts,
                                                   C.byref(seen), C.byref(kept)))
        return seen.value, kept.value

    def pairs_append(self, seq1: np.ndarray, off1: np.ndarray, seq2: np.ndarray, off2: np.ndarray,
                     count_mate2: Optional[np.ndarray] = None, flags: Optional[np.ndarray] = None):
        """flags (one byte per pair: 1 = mate 1 counted in phase A, 2 = mate 2 counted, 4 = pair voted in phase C) overrides
        count_mate2 (0 = mate 2 not counted, the fq2-cut of E:1438-1445)"""
        s1 = np.ascontiguousarray(seq1, dtype=np.uint8)
        s2 = np.ascontiguousarray(seq2, dtype=np.uint8)
        o1 = np.ascontiguousarray(off1, dtype=np.uint64)
        o2 = np.ascontiguousarray(off2, dtype=np.uint64)
        assert o1.size == o2.size
        if flags is not None:
            fl = np.ascontiguousarray(flags, dtype=np.uint8)
            assert fl.size == o1.size - 1
            _lib.check(self.lib.lhgt_pairs_append_flags(self.h, _ptr(s1, C.c_uint8), _ptr(o1, C.c_uint64), _ptr(s2, C.c_uint8),
                                                        _ptr(o2, C.c_uint64), o1.size - 1, _ptr(fl, C.c_uint8)))
            return
        c2 = None if count_mate2 is None else np.ascontiguousarray(count_mate2, dtype=np.uint8)
        _lib.check(self.lib.lhgt_pairs_append(self.h, _ptr(s1, C.c_uint8), _ptr(o1, C.c_uint64), _ptr(s2, C.c_uint8),
                                              _ptr(o2, C.c_uint64), o1.size - 1,
                                              None if c2 is None else _ptr(c2, C.c_uint8)))

    def set_count_on_load(self, on: bool):
        """pairs_load_fastq then runs phase A on every batch while the next one is parsed (needs the coder: index first)"""
        _lib.check(self.lib.lhgt_set_count_on_load(self.h, 1 if on else 0))

    def pairs_clear(self):
        _lib.check(self.lib.lhgt_pairs_clear(self.h))

    def pairs_count(self) -> int:
        n = C.c_long(0)
        _lib.check(self.lib.lhgt_pairs_count(self.h, C.byref(n)))
        return n.value

    # ---- synthetic workload (bench / tests)
    def synth_reference(self, ref_seed: int, n_contigs: int, contig_len: int, want_host: bool = False):
        host = np.zeros(n_contigs * contig_len, dtype=np.uint8) if want_host else None
        _lib.check(self.lib.lhgt_synth_reference(self.h, ref_seed, n_contigs, contig_len,
                                                 None if host is None else _ptr(host, C.c_uint8)))
        return host

    def synth_reference_shard(self, ref_seed: int, n_contigs: int, contig_len: int, rank: int, world: int, want_host: bool = False):
        """contigs [n_contigs*rank/world, n_contigs*(rank+1)/world) of the synthetic reference become the resident shard"""
        n = n_contigs * (rank + 1) // world - n_contigs * rank // world
        host = np.zeros(n * contig_len, dtype=np.uint8) if want_host else None
        _lib.check(self.lib.lhgt_synth_reference_shard(self.h, ref_seed, n_contigs, contig_len, rank, world,
                                                       None if host is None else _ptr(host, C.c_uint8)))
        return host

    def synth_reference_cuts(self, ref_seed: int, n_contigs: int, contig_len: int, cuts: np.ndarray, want_host: bool = False):
        """the same base stream as synth_reference, cut into contigs at `cuts` (ascending, 0 .. n_contigs*contig_len)"""
        cu = np.ascontiguousarray(cuts, dtype=np.uint64)
        host = np.zeros(n_contigs * contig_len, dtype=np.uint8) if want_host else None
        _lib.check(self.lib.lhgt_synth_reference_cuts(self.h, ref_seed, n_contigs, contig_len, _ptr(cu, C.c_uint64), cu.size,
                                                      None if host is None else _ptr(host, C.c_uint8)))
        return host

    def synth_options(self, snp_permille: int = 0, n_permille: int = 20, sample_contigs: int = 0):
        _lib.check(self.lib.lhgt_synth_options(self.h, snp_permille, n_permille, sample_contigs))

    def synth_read_mix(self, long_permille: int, long_len: int):
        """long_permille of 1000 synthetic pairs get reads of long_len bases (0: all of synth_pairs' read_len)"""
        _lib.check(self.lib.lhgt_synth_read_mix(self.h, long_permille, long_len))

    def synth_pairs(self, ref_seed: int, reads_seed: int, n_contigs: int, contig_len: int, first_pair: int,
                    n_pairs: int, read_len: int = 150, want_host: bool = False):
        h1 = np.zeros(n_pairs * read_len, dtype=np.uint8) if want_host else None
        h2 = np.zeros(n_pairs * read_len, dtype=np.uint8) if want_host else None
        _lib.check(self.lib.lhgt_synth_pairs(self.h, ref_seed, reads_seed, n_contigs, contig_len, first_pair, n_pairs,
                                             read_len, None if h1 is None else _ptr(h1, C.c_uint8),
                                             None if h2 is None else _ptr(h2, C.c_uint8)))
        return h1, h2

    # ---- phases
    def count_kmers(self):
        _lib.check(self.lib.lhgt_count_kmers(self.h))

    def set_count_mode(self, mode: int):
        _lib.check(self.lib.lhgt_set_count_mode(self.h, mode))

    # ---- count_diff_kmer.cpp compatibility (include/localhgt_hip.h)
    def set_count_compat(self, on: bool):
        _lib.check(self.lib.lhgt_set_count_compat(self.h, 1 if on else 0))

    def coder_generate_count_diff(self):
        _lib.check(self.lib.lhgt_coder_generate_count_diff(self.h))

    def reads_load_count_diff(self, fq: str, size_for_chunks: int, ratio_percent: int, seed: int) -> int:
        n = C.c_long(0)
        _lib.check(self.lib.lhgt_reads_load_count_diff(self.h, fq.encode(), size_for_chunks, int(ratio_percent), C.c_uint(seed & 0xFFFFFFFF), C.byref(n)))
        return n.value

    def counts_clear(self):
        _lib.check(self.lib.lhgt_counts_clear(self.h))

    def ref_scan(self, hit_ratio: float, match_ratio: float, max_peak: int) -> int:
        n = C.c_long(0)
        _lib.check(self.lib.lhgt_ref_scan(self.h, C.c_float(np.float32(hit_ratio)), C.c_float(np.float32(match_ratio)),
                                          int(max_peak), C.byref(n)))
        return n.value

    # ---- reference-sharded phase B
    def ref_scan_local(self, hit_ratio: float, match_ratio: float) -> Tuple[int, int]:
        n, s = C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_ref_scan_local(self.h, C.c_float(np.float32(hit_ratio)), C.c_float(np.float32(match_ratio)),
                                                C.byref(n), C.byref(s)))
        return n.value, s.value

    def ref_scan_group_counts(self) -> list:
        """-t N emulation on a reference shard: new peaks of the local contigs per split_ref group"""
        n = self.emulated_threads
        out = np.zeros(n, dtype=np.int64)
        _lib.check(self.lib.lhgt_ref_scan_group_counts(self.h, _ptr(out, C.c_long), n))
        return [int(x) for x in out]

    def set_group_totals(self, totals, max_peak: int) -> int:
        t = np.ascontiguousarray(totals, dtype=np.int64)
        first = C.c_long(0)
        _lib.check(self.lib.lhgt_set_group_totals(self.h, _ptr(t, C.c_long), t.size, int(max_peak), C.byref(first)))
        return first.value

    def ref_scan_emit(self, id_base: int) -> Tuple[int, int, int]:
        """(device ptr of int32 loci[2*n_new_local], device ptr of uint32 regs[2*n_regs], n_regs)"""
        pl, pr, n = C.c_void_p(), C.c_void_p(), C.c_long(0)
        _lib.check(self.lib.lhgt_ref_scan_emit(self.h, id_base, C.byref(pl), C.byref(pr), C.byref(n)))
        return pl.value or 0, pr.value or 0, n.value

    def peaks_install(self, n_peaks_total: int, n_selected_total: int, max_peak: int, loci_ptr: int, regs_ptr: int, n_regs: int):
        _lib.check(self.lib.lhgt_peaks_install(self.h, n_peaks_total, n_selected_total, int(max_peak), C.c_void_p(loci_ptr),
                                               C.c_void_p(regs_ptr), n_regs))

    def vote(self):
        _lib.check(self.lib.lhgt_vote(self.h))

    def write_intervals(self, path: str) -> int:
        n = C.c_long(0)
        _lib.check(self.lib.lhgt_write_intervals(self.h, path.encode(), C.byref(n)))
        return n.value

    def set_thread_emulation(self, threads: int):
        """the reference's -t N without its races (include/localhgt_hip.h: lhgt_set_thread_emulation); 1 = off"""
        _lib.check(self.lib.lhgt_set_thread_emulation(self.h, int(threads)))
        self.emulated_threads = int(threads)

    def set_cu_mask(self, cus=None, n_cus: int = 256):
        """run this context's kernels only on the CUs listed (None: all)"""
        if cus is None:
            _lib.check(self.lib.lhgt_set_cu_mask(self.h, None, 0))
            return
        m = np.zeros((n_cus + 31) // 32, dtype=np.uint32)
        for c in cus:
            m[c >> 5] |= np.uint32(1 << (c & 31))
        _lib.check(self.lib.lhgt_set_cu_mask(self.h, _ptr(m, C.c_uint32), m.size))

    def set_debug(self, flags: int):
        _lib.check(self.lib.lhgt_set_debug(self.h, flags))

    def phase_ms(self, phase: int) -> float:
        ms = C.c_float(0)
        _lib.check(self.lib.lhgt_phase_ms(self.h, phase, C.byref(ms)))
        return ms.value

    def scan_info(self) -> dict:
        """form of the last ref_scan: {'lite': bool, 'frac_slots_at_3': float, 'tiles': int, 'tiles_exact': int}"""
        lite, frac, nt, ne = C.c_int(0), C.c_double(0), C.c_long(0), C.c_long(0)
        _lib.check(self.lib.lhgt_scan_info(self.h, C.byref(lite), C.byref(frac), C.byref(nt), C.byref(ne)))
        return {"lite": lite.value in (1, 4), "form": ("exact", "single-first", "trio-first", "slot-first", "slot-single")[lite.value], "frac_slots_at_3": round(frac.value, 4),
                "tiles": nt.value, "tiles_exact": ne.value}

    def registry_info(self) -> dict:
        """how the last ref_scan registered its peaks' k-mers (include/localhgt_hip.h: lhgt_registry_info): chunks 0 = the direct kernel"""
        ch, bound, direct = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.lhgt_registry_info(self.h, C.byref(ch), C.byref(bound), C.byref(direct)))
        return {"chunks": ch.value, "records_bound": bound.value, "records_direct": direct.value}

    def slot_list(self, mode: int = -1) -> dict:
        """the slot list of the resident reference (include/localhgt_hip.h: lhgt_slot_list): mode 0 never / drop, 1 before the second
        sparse-form scan of a reference (default), 2 before the first, -1 query"""
        n, b = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.lhgt_slot_list(self.h, int(mode), C.byref(n), C.byref(b)))
        ms = C.c_double(0)
        _lib.check(self.lib.lhgt_slot_list_build_ms(self.h, C.byref(ms)))
        return {"entries": n.value, "bytes": b.value, "build_ms": round(ms.value, 1)}

    WORK_STATS = ("count_keys", "scan_probes", "scan_followed", "vote_l2_probes", "vote_hbm_probes", "vote_revoted_pairs", "vote_shared_fetches", "vote_shared_outside")

    def work_stats(self, enable: int = -1) -> dict:
        """work counters of the phases run since work_stats(1) (include/localhgt_hip.h: lhgt_work_stats); measurement only"""
        out = np.zeros(8, dtype=np.uint64)
        _lib.check(self.lib.lhgt_work_stats(self.h, int(enable), _ptr(out, C.c_uint64)))
        return {name: int(v) for name, v in zip(self.WORK_STATS, out) if name}

    def vote_info(self) -> dict:
        """which kernel the last vote() took and the size of its bitmap (include/localhgt_hip.h: lhgt_vote_info)"""
        f, b, q = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self.lib.lhgt_vote_info(self.h, C.byref(f), C.byref(b), C.byref(q)))
        mib = (1 << b.value) / 8 / (1 << 20) * (0.75 if q.value else 1.0) if b.value else 0.0
        return {"form": ("dense", "bitmap", "queued", "fold", "shared")[f.value], "bitmap_MiB": round(mib, 3)}

    def synchronize(self):
        _lib.check(self.lib.lhgt_synchronize(self.h))

    # ---- device buffers for the multi-GPU exchange
    def counts_buffer(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t(0)
        _lib.check(self.lib.lhgt_counts_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def counts_merge(self, dev_ptr: int, byte_offset: int, nbytes: int):
        _lib.check(self.lib.lhgt_counts_merge(self.h, C.c_void_p(dev_ptr), byte_offset, nbytes))

    def filter_buffer(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t(0)
        _lib.check(self.lib.lhgt_filter_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- introspection (parity tests)
    DIGEST_COUNTS, DIGEST_FLAGS, DIGEST_PEAK_KMER, DIGEST_LOCI, DIGEST_VOTES = range(5)

    def digest(self, what: int, mask: int = 0xFFFFFFFFFFFFFFFF) -> Tuple[int, int]:
        """(position-sensitive checksum, non-zero entries) of a whole device table"""
        out = np.zeros(2, dtype=np.uint64)
        _lib.check(self.lib.lhgt_digest(self.h, what, C.c_uint64(mask), _ptr(out, C.c_uint64)))
        return int(out[0]), int(out[1])

    def counts_export(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = (1 << self.k) - first if n is None else n
        out = np.zeros(n, dtype=np.uint8)
        _lib.check(self.lib.lhgt_counts_export_u8(self.h, first, n, _ptr(out, C.c_uint8)))
        return out

    def counts_histogram(self) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint64)
        _lib.check(self.lib.lhgt_counts_histogram(self.h, _ptr(out, C.c_uint64)))
        return out

    def flags_export(self, first: int, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint8)
        _lib.check(self.lib.lhgt_flags_export(self.h, first, n, _ptr(out, C.c_uint8)))
        return out

    def peaks_export(self, n: int) -> Tuple[np.ndarray, np.ndarray]:
        loci = np.zeros(2 * max(n, 1), dtype=np.int32)
        filt = np.zeros(max(n, 1), dtype=np.uint8)
        _lib.check(self.lib.lhgt_peaks_export(self.h, _ptr(loci, C.c_int32), _ptr(filt, C.c_uint8), n))
        return loci[:2 * n], filt[:n]

    def vote_groups_export(self) -> Optional[np.ndarray]:
        """the group bounds of the dense vote's bound (include/localhgt_hip.h: lhgt_vote_groups_export), or None if the last scan left none"""
        out = np.zeros(1025, dtype=np.uint32)
        ok = C.c_int(0)
        _lib.check(self.lib.lhgt_vote_groups_export(self.h, _ptr(out, C.c_uint32), 1025, C.byref(ok)))
        return out if ok.value else None

    def peak_kmer_export(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = (1 << self.k) - first if n is None else n
        out = np.zeros(n, dtype=np.uint32)
        _lib.check(self.lib.lhgt_peak_kmer_export(self.h, first, n, _ptr(out, C.c_uint32)))
        return out
