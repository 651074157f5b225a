"""A sample kept packed on disk (round 6): `localhgt_pack fq1 fq2 out.lhgp`, and the header the loader reads back.

The reference reads FASTQ text (src/extract_ref_normal_peak.cpp:1020-1044, 350-419) and so does `extract_ref`; from text one GPU's
share of a host parses 86-100 M pairs/s and N ranks on one host share that rate -- the host side is at copy speed, 637 bytes per
pair.  A packed sample is the resident read store's own records (2-bit planes + a not-a-base plane per mate, 148 bytes per 150-base
pair at a fixed stride) behind a header with everything the loader decides from the TEXT: the lines each thread of the reference's
`-t N` consumes for every N up to --max-threads (get_fq_start and the chunk loops, E:44-89, 1019-1026: computed from the text here),
the first pair whose mate 2 lies behind size(fq1) (quirk Q4, E:1419-1445), the bases of fq1 (cal_sam_ratio, E:1244-1270).  Which
reads a run keeps is decided at load time from those tables (csrc/k_packed.hip), so `extract_ref S.lhgp - ref.fa ...` writes the
files `extract_ref S.1.fq S.2.fq ref.fa ...` writes -- for every seed, --sample and -t.  Only record-aligned pairs of files are
packed (same number of records, same first read ID, no line beyond the reference's 500-character buffers); for everything else
the FASTQ loader is the way."""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import sys
from typing import Optional

import numpy as np

MAGIC = b"LHGTPK01"
DATA_OFFSET = 1 << 20          # the records start here; the header (JSON) lies in front of them


def is_packed(path: str) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(8) == MAGIC
    except OSError:
        return False


class Header:
    def __init__(self, d: dict, path: str):
        self.d, self.path = d, path
        self.n_pairs = int(d["n_pairs"])
        self.stride = int(d["stride"])
        self.q4_first_pair = int(d["q4_first_pair"])
        self.fq1_bases = int(d["fq1_bases"])
        self.data_offset = int(d["data_offset"])

    def ratio(self, sample: float) -> float:
        """cal_sam_ratio and the <= 1 branch (E:1392-1398, 1244-1270)"""
        if sample <= 1:
            return 100.0 * sample
        if self.fq1_bases == 0:
            return float("inf")
        return 100.0 * sample / (2.0 * self.fq1_bases)

    def thread_chunks(self, threads: int):
        """(first1, count1, first2, count2) of the reference's -t threads on the packed files, or raises the refusal the FASTQ loader
        would have raised (LocalHGTError 9: only the emulation refuses; the caller falls back to -t 1)"""
        from ._lib import LocalHGTError
        t = self.d["threads"].get(str(threads))
        if t is None:
            raise LocalHGTError(9, f"-t {threads} emulation: {self.path} was packed with the thread chunks of -t 2 .. {self.d['max_threads']} only "
                                   f"(localhgt_pack --max-threads {threads})")
        if "refused" in t:
            raise LocalHGTError(int(t["refused"][0]), t["refused"][1])
        return tuple(np.asarray(t[k], dtype=np.int64) for k in ("first1", "count1", "first2", "count2"))


def read_header(path: str) -> Header:
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise SystemExit(f"{path}: not a packed sample (localhgt_pack)")
        (n,) = struct.unpack("<Q", f.read(8))
        d = json.loads(f.read(n).decode())
    if d.get("version") != 1:
        raise SystemExit(f"{path}: packed sample of version {d.get('version')}, this build reads version 1")
    return Header(d, path)


def _first_id(path: str) -> bytes:
    """get_read_ID of the file's first line (E:303-311): cut at the first '/', then ' ', then tab"""
    with open(path, "rb") as f:
        line = f.readline().rstrip(b"\n")
    for sep in (b"/", b" ", b"\t"):
        i = line.find(sep)
        if i >= 0:
            line = line[:i]
    return line


def _lines(lib, path: str) -> int:
    """the file's lines (lhgt_fastq_plan_part: host only)"""
    n, tot = C.c_long(0), C.c_long(0)
    from . import _lib
    _lib.check(lib.lhgt_fastq_plan_part(path.encode(), lib.lhgt_fastq_plan_chunk_bytes(), 0, 1, None, None, 0, C.byref(n), C.byref(tot), None))
    st, cn = np.zeros(max(n.value, 1), dtype=np.uint64), np.zeros(max(n.value, 1), dtype=np.int64)
    _lib.check(lib.lhgt_fastq_plan_part(path.encode(), lib.lhgt_fastq_plan_chunk_bytes(), 0, 1, st.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        cn.ctypes.data_as(C.POINTER(C.c_long)), n.value, C.byref(n), C.byref(tot), None))
    return int(cn[:n.value].sum())


def pack(fq1: str, fq2: str, out: str, max_threads: int = 32, device: int = 0, log=print, host: Optional[bool] = None) -> dict:
    """host=True: no GPU is touched -- the loader's host parse and a host restatement of the 2-bit packing write the same bytes
    (lhgt_fastq_pack_host; round 6, late); host=False: through the resident store on the GPU; None: the GPU if there is one"""
    from . import _lib
    lib = _lib.load(require_gpu=False)
    if _first_id(fq1) != _first_id(fq2):
        raise SystemExit(f"{fq1} and {fq2} open with different read IDs: the reference re-synchronises such files by ID (E:368-402); keep them as FASTQ")
    if host is None:
        host = os.environ.get("LHGT_PACK_HOST", "") == "1" or not _lib.gpu_available()
    if host:
        lines1, lines2 = _lines(lib, fq1), _lines(lib, fq2)
        if lines1 != lines2 or lines1 % 4:
            raise SystemExit(f"{fq1} has {lines1} lines, {fq2} {lines2}: only record-aligned pairs of files are packed; keep these as FASTQ")
        with open(out, "wb") as f:
            f.truncate(DATA_OFFSET)
        n, q4, bases, stride_c, ml = C.c_long(0), C.c_long(0), C.c_uint64(0), C.c_long(0), C.c_int(0)
        _lib.check(lib.lhgt_fastq_pack_host(fq1.encode(), fq2.encode(), out.encode(), DATA_OFFSET, 0, C.byref(stride_c), C.byref(n), C.byref(q4),
                                            C.byref(bases), C.byref(ml)))
        if n.value != lines1 // 4:
            raise SystemExit(f"the loader kept {n.value} pairs of {lines1 // 4} records: not a clean pair of files; keep them as FASTQ")
        stride, max_len = stride_c.value, ml.value
        return _finish(lib, fq1, fq2, out, max_threads, n, q4, bases, stride, max_len, lines1, log)
    from .engine import Engine
    with Engine(32, 3, device) as eng:
        p1, p2 = eng.fastq_plan(fq1, False, other=fq2)
        lines1, lines2 = int(p1[1].sum()), int(p2[1].sum())
        if lines1 != lines2 or lines1 % 4:
            raise SystemExit(f"{fq1} has {lines1} lines, {fq2} {lines2}: only record-aligned pairs of files are packed; keep these as FASTQ")
        eng.rng_seed(1)
        eng.sampling_init(100.0)
        eng.set_thread_emulation(1)
        seen, kept = eng.pairs_load_fastq(fq1, fq2, 100.0)
        if not (seen == kept == lines1 // 4):
            raise SystemExit(f"the loader kept {kept} of {seen} pairs of {lines1 // 4} records: not a clean pair of files; keep them as FASTQ")
        nb = C.c_long(0)
        _lib.check(lib.lhgt_pairs_batches(eng.h, C.byref(nb)))
        max_len = 0
        for b in range(nb.value):
            ml = C.c_int(0)
            _lib.check(lib.lhgt_pairs_batch_info(eng.h, b, None, None, C.byref(ml)))
            max_len = max(max_len, ml.value)
        stride = 4 + 24 * ((max_len + 31) // 32 + 1)
        with open(out, "wb") as f:
            f.truncate(DATA_OFFSET)
        n, q4, bases = C.c_long(0), C.c_long(0), C.c_uint64(0)
        _lib.check(lib.lhgt_pairs_store_write(eng.h, out.encode(), DATA_OFFSET, stride, C.byref(n), C.byref(q4), C.byref(bases)))
    return _finish(lib, fq1, fq2, out, max_threads, n, q4, bases, stride, max_len, lines1, log)


def _finish(lib, fq1, fq2, out, max_threads, n, q4, bases, stride, max_len, lines1, log) -> dict:
    """the header: what the loader decides from the text, for every -t N up to max_threads"""
    from . import _lib
    size1 = os.path.getsize(fq1)
    threads = {}
    for t in range(2, max_threads + 1):
        ent = {}
        try:
            for tag, fq in (("1", fq1), ("2", fq2)):
                eb, fl, nl = (np.zeros(t, dtype=np.int64) for _ in range(3))
                _lib.check(lib.lhgt_fastq_thread_chunks(fq.encode(), size1, t, eb.ctypes.data_as(C.POINTER(C.c_long)),
                                                        fl.ctypes.data_as(C.POINTER(C.c_long)), nl.ctypes.data_as(C.POINTER(C.c_long))))
                ent["first" + tag], ent["count" + tag] = [int(x) for x in fl], [int(x) for x in nl]
        except _lib.LocalHGTError as ex:
            if ex.code != 9:
                raise
            msg = str(ex).split(": ", 2)[-1]
            ent = {"refused": [9, msg]}
        threads[str(t)] = ent
    st1, st2 = os.stat(fq1), os.stat(fq2)
    hdr = {"version": 1, "n_pairs": n.value, "stride": stride, "max_len": max_len, "q4_first_pair": q4.value, "fq1_bases": bases.value,
           "data_offset": DATA_OFFSET, "lines": lines1, "max_threads": max_threads, "threads": threads,
           "sources": {"fq1": {"name": os.path.basename(fq1), "size": st1.st_size, "mtime_ns": st1.st_mtime_ns},
                       "fq2": {"name": os.path.basename(fq2), "size": st2.st_size, "mtime_ns": st2.st_mtime_ns}}}
    blob = json.dumps(hdr).encode()
    if 16 + len(blob) > DATA_OFFSET:
        raise SystemExit("the header does not fit in front of the records: fewer --max-threads")
    with open(out, "r+b") as f:
        f.write(MAGIC + struct.pack("<Q", len(blob)) + blob)
    log(f"{out}: {n.value} pairs, {stride} bytes each ({(st1.st_size + st2.st_size) / max(1, os.path.getsize(out)):.1f} x smaller than the text), "
        f"thread chunks for -t 2 .. {max_threads}" + (f", mate 2 uncounted from pair {q4.value} on (quirk Q4)" if q4.value < n.value else ""))
    return hdr


def main(argv: Optional[list] = None) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="localhgt_pack", description="pack a record-aligned pair of FASTQ files for extract_ref (give the result as fq1 and '-' as fq2)")
    ap.add_argument("fq1")
    ap.add_argument("fq2")
    ap.add_argument("out")
    ap.add_argument("--max-threads", type=int, default=32, help="thread chunks of the reference's -t 2 .. N are stored (localhgt bkp passes -t 10)")
    ap.add_argument("--host", action="store_true", help="pack on the host's CPUs, no GPU touched (the default where there is none; same bytes)")
    ap.add_argument("--gpu", action="store_true", help="pack through the resident store on the GPU")
    a = ap.parse_args(argv)
    if not 1 <= a.max_threads <= 99:
        raise SystemExit("--max-threads: 1 .. 99 (split_ref holds 100 groups)")
    pack(a.fq1, a.fq2, a.out, a.max_threads, host=True if a.host else (False if a.gpu else None))
    return 0


if __name__ == "__main__":
    sys.exit(main())
