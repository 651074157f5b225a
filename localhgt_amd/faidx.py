"""BED -> extracted FASTA and the .fai index, without samtools.

Host-side mirror of two command lines of the reference:
  samtools faidx -r ${interval_file}.bed $original_ref > $extracted_ref      scripts/pipeline.sh:37
  samtools faidx $ref                                                        scripts/infer_HGT_breakpoint.py:156
The work is done by the C-ABI (lhgt_faidx_extract / lhgt_faidx_build, csrc/host_faidx.cpp); nothing here touches a GPU.
samtools is not installed in the image, so the output format follows its documented behaviour and parity is unpinned."""
from __future__ import annotations

import ctypes as C
import sys

from . import _lib


def build_fai(fasta: str, fai: str | None = "") -> int:
    """write <fasta>.fai (fai="" -> default path, None -> validate only); returns the number of sequences"""
    lib = _lib.load(require_gpu=False)
    n = C.c_long()
    path = None if fai is None else (fai or fasta + ".fai").encode()
    _lib.check(lib.lhgt_faidx_build(fasta.encode(), path, C.byref(n)))
    return n.value


def extract_regions(fasta: str, regions: str, out: str = "-", line_width: int = 60):
    """one FASTA record per line of `regions` (NAME:BEG-END, 1-based inclusive); returns (regions, bases)"""
    lib = _lib.load(require_gpu=False)
    n_reg, n_bases = C.c_long(), C.c_long()
    if out == "-":
        sys.stdout.flush()
    _lib.check(lib.lhgt_faidx_extract(fasta.encode(), regions.encode(), out.encode(), line_width, C.byref(n_reg), C.byref(n_bases)))
    return n_reg.value, n_bases.value


def main(argv=None) -> int:
    """`faidx [-n WIDTH] [-o OUT] [-r REGION_FILE] <ref.fa>` -- the two forms pipeline.sh uses"""
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "faidx":
        argv = argv[1:]
    regions, out, width, pos = None, "-", 60, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ("-r", "--region-file"):
            regions = argv[i + 1]; i += 2
        elif a in ("-o", "--output"):
            out = argv[i + 1]; i += 2
        elif a in ("-n", "--length"):
            width = int(argv[i + 1]); i += 2
        else:
            pos.append(a); i += 1
    if len(pos) != 1:
        print("usage: localhgt_faidx [faidx] [-n WIDTH] [-o OUT] [-r REGION_FILE] <ref.fa>", file=sys.stderr)
        return 1
    try:
        if regions is None:
            build_fai(pos[0])
        else:
            extract_regions(pos[0], regions, out, width)
    except _lib.LocalHGTError as err:
        print(f"[faidx] {err}", file=sys.stderr)
        return 1
    return 0
