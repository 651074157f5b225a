"""ctypes view of oracle/liblhgt_oracle.so -- the CPU checker. Imported by tests/ and by
bench.py's cpu_baseline leg only; nothing under localhgt_amd/ may import it."""
import ctypes as C

import numpy as np


class Report(C.Structure):
    _fields_ = [("t_index", C.c_double), ("t_count", C.c_double), ("t_scan", C.c_double), ("t_vote", C.c_double),
                ("t_total", C.c_double), ("pairs_counted", C.c_long), ("pairs_voted", C.c_long),
                ("n_peaks", C.c_long), ("n_filtered", C.c_long)]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, so_path):
        L = self.L = C.CDLL(so_path)
        L.orc_hash_kmer.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_short), C.POINTER(C.c_uint32)]
        L.orc_random_coder.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_short)]
        L.orc_sampling_array.restype = C.POINTER(C.c_float)
        L.orc_sampling_array.argtypes = [C.c_long]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_short)]
        L.orc_index_header.argtypes = [C.c_char_p, C.POINTER(C.c_short)]
        L.orc_sam_ratio.restype = C.c_double
        L.orc_sam_ratio.argtypes = [C.c_char_p, C.c_double]
        L.orc_file_size.restype = C.c_long
        L.orc_file_size.argtypes = [C.c_char_p]
        L.orc_count_fastq.restype = C.c_long
        L.orc_count_fastq.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int, C.POINTER(C.c_short), C.c_double,
                                      C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_int]
        L.orc_ref_scan.restype = C.c_long
        L.orc_ref_scan.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_float, C.c_float, C.c_long,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8),
                                   C.POINTER(C.c_long)]
        L.orc_vote.restype = C.c_long
        L.orc_vote.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_short), C.c_double,
                               C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                               C.POINTER(C.c_uint8), C.c_int]
        L.orc_write_intervals.restype = C.c_long
        L.orc_write_intervals.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_long]
        L.orc_get_fq_start.restype = C.c_long
        L.orc_get_fq_start.argtypes = [C.c_char_p, C.c_long, C.c_long]
        L.orc_run_threads.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                      C.c_long, C.c_int, C.c_uint, C.c_double, C.POINTER(Report)]
        L.orc_count_diff_kmer.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_uint64)]
        L.orc_run.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_int,
                              C.c_long, C.c_int, C.c_uint, C.c_double, C.POINTER(Report)]

    # ---- pieces
    def set_pretouch(self, on: bool):
        self.L.orc_set_pretouch(1 if on else 0)

    def srand(self, seed):
        self.L.orc_srand(C.c_uint(seed))

    def random_coder(self, k, e):
        cc = np.zeros(300, dtype=np.int16)
        self.L.orc_random_coder(k, e, _p(cc, C.c_short))
        return cc

    def sampling_array(self, n):
        ptr = self.L.orc_sampling_array(n)
        out = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        self.L.orc_free(ptr)
        return out

    def hash_kmer(self, kmer: bytes, k, e, cc):
        out = np.zeros(16, dtype=np.uint32)
        ok = self.L.orc_hash_kmer(kmer, k, e, _p(cc, C.c_short), _p(out, C.c_uint32))
        return bool(ok), out[:e].copy()

    def index_build(self, fasta, index_path, len_path, k, e, cc):
        return self.L.orc_index_build(fasta.encode(), index_path.encode(), len_path.encode(), k, e, _p(cc, C.c_short))

    def index_header(self, index_path):
        cc = np.zeros(300, dtype=np.int16)
        rc = self.L.orc_index_header(index_path.encode(), _p(cc, C.c_short))
        assert rc == 0
        return cc

    def sam_ratio(self, fq1, sample):
        return self.L.orc_sam_ratio(fq1.encode(), sample)

    def file_size(self, path):
        return self.L.orc_file_size(path.encode())

    def count(self, fq, byte_limit, k, e, cc, ratio, rnd, table, threads=1):
        rp = _p(rnd, C.c_float) if rnd is not None else None
        return self.L.orc_count_fastq(fq.encode(), byte_limit, k, e, _p(cc, C.c_short), ratio, rp,
                                      _p(table, C.c_uint8), threads)

    def ref_scan(self, index_path, table, k, e, hit_ratio, match_ratio, max_peak, peak_kmer, flags=None):
        loci = np.zeros(2 * max_peak, dtype=np.int32)
        ext = C.c_long(0)
        n = self.L.orc_ref_scan(index_path.encode(), _p(table, C.c_uint8), k, e, hit_ratio, match_ratio, max_peak,
                                _p(loci, C.c_int32), _p(peak_kmer, C.c_uint32),
                                _p(flags, C.c_uint8) if flags is not None else None, C.byref(ext))
        return n, loci, ext.value

    def vote(self, fq1, fq2, k, e, cc, ratio, rnd, peak_kmer, loci, n_peaks, threads=1):
        pf = np.zeros(max(n_peaks, 1), dtype=np.uint8)
        rp = _p(rnd, C.c_float) if rnd is not None else None
        kept = self.L.orc_vote(fq1.encode(), fq2.encode(), k, e, _p(cc, C.c_short), ratio, rp,
                               _p(peak_kmer, C.c_uint32), _p(loci, C.c_int32), _p(pf, C.c_uint8), threads)
        return kept, pf

    def write_intervals(self, path, loci, pf, n_peaks):
        return self.L.orc_write_intervals(path.encode(), _p(loci, C.c_int32), _p(pf, C.c_uint8), n_peaks)

    def get_fq_start(self, data: bytes, start: int) -> int:
        return self.L.orc_get_fq_start(data, len(data), start)

    def run_threads(self, fq1, fq2, fasta, interval, hit_ratio, match_ratio, threads, k, max_peak, e, seed, sample):
        """the reference's -t N without its races (threads one after the other in creation order)"""
        rep = Report()
        rc = self.L.orc_run_threads(fq1.encode(), fq2.encode(), fasta.encode(), interval.encode(), hit_ratio, match_ratio,
                                    threads, k, max_peak, e, seed, sample, C.byref(rep))
        return rc, rep

    def count_diff_kmer(self, fq1, fq2, k, ratio, time_seed=1):
        """the reference's stand-alone phase-A tool with time() fixed and its threads in creation order; returns (rc, hist[4])"""
        hist = np.zeros(4, dtype=np.uint64)
        rc = self.L.orc_count_diff_kmer(fq1.encode(), fq2.encode(), k, int(ratio), time_seed, _p(hist, C.c_uint64))
        return rc, hist

    # ---- whole run with the 12-argument contract
    def run(self, fq1, fq2, fasta, interval, hit_ratio, match_ratio, threads, k, max_peak, e, seed, sample):
        rep = Report()
        rc = self.L.orc_run(fq1.encode(), fq2.encode(), fasta.encode(), interval.encode(), hit_ratio, match_ratio,
                            threads, k, max_peak, e, seed, sample, C.byref(rep))
        return rc, rep
