"""Seeded parity cases shared by tests/golden/make_golden.py and the test-suite.

Each case is generated from seeds by localhgt_amd.synth (inputs are not committed, their
sha256 is, so a drift of the generator is detected rather than silently accepted) and run
through the 12-argument `extract_ref` contract
(/root/reference/src/extract_ref_normal_peak.cpp:1352-1364).
"""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

from localhgt_amd import synth


@dataclass
class Case:
    name: str
    k: int = 24
    e: int = 3
    seed: int = 1
    sample: float = 1
    hit_ratio: float = 0.1
    match_ratio: float = 0.08
    max_peak: int = 1000000
    # generator knobs
    ref_seed: int = 11
    n_contigs: int = 8
    min_len: int = 20000
    max_len: int = 40000
    short_contig_at: Optional[int] = 8          # index in the contig list (None = no short contig)
    n_run_at: Optional[Tuple[int, int, int]] = None
    reads_seed: int = 12
    depth: float = 12
    snp_rate: float = 0.0
    n_read_frac: float = 0.02
    fq2_header_pad: int = 0
    fq1_header_pad: int = 0
    fq1_drop_tail: int = 0                      # fq1 has that many records fewer than fq2
    fq1_trailing_blank: bool = False
    fq2_stray_records: int = 0                  # foreign records in front of fq2: phase C re-scans fq2 for fq1's first read ID (E:368-402)
    fq2_drop_tail: int = 0                      # fq2 has that many records fewer than fq1 (E:356-367)
    fq2_last_line_bases_of: Optional[int] = None  # fq2's last line holds the bases of that read's mate 2 and has no newline
    long_line: Optional[Tuple[int, int, int]] = None   # (file, read, length): a line longer than the reference's buffers (harmless while unsampled)
    threads: int = 1                            # -t of the run (parity contract of t > 1: oracle/_ref run on ONE core)
    lowercase_every: int = 0
    preexisting_index: bool = False             # run twice, keep outputs of the 2nd run (quirk Q3)
    bed_defined: bool = True                    # False when a short contig sits mid-file (quirk Q7)
    notes: str = ""


CASES: Dict[str, Case] = {c.name: c for c in [
    Case("k24_base", notes="short contig last; every read kept"),
    Case("k32_base", k=32, notes="default k; 2^32 tables"),
    Case("k21_e3", k=21, ref_seed=21, reads_seed=22, snp_rate=0.01, notes="config-5 k; 1% SNPs"),
    Case("k24_sample_half_fresh", sample=0.5, notes="sampling active, index built in-run (Q3)"),
    Case("k24_sample_half_cached", sample=0.5, preexisting_index=True, notes="sampling active, index cached (Q3)"),
    Case("k24_sample_bases", sample=700000.0, notes="--sample > 1: cal_sam_ratio path"),
    Case("k24_fq2_longer", fq2_header_pad=20, notes="fq2 bigger than fq1: mate-2 counting truncated (Q4)"),
    Case("k24_short_mid", short_contig_at=2, bed_defined=False, notes="short contig mid-file (Q7): interval only"),
    Case("k20_e2", k=20, e=2, ref_seed=31, reads_seed=32, notes="two hashes"),
    Case("k22_e5", k=22, e=5, ref_seed=41, reads_seed=42, match_ratio=0.04, notes="five hashes: two rand() rows per position"),
    Case("k24_nrun_lower", n_run_at=(1, 9000, 40), lowercase_every=7, snp_rate=0.005, ref_seed=51, reads_seed=52,
         notes="N run in the reference (Q6), lower-case read bases, SNPs"),
    Case("k24_fq2_surplus", fq1_header_pad=30, fq1_drop_tail=300, fq1_trailing_blank=True, ref_seed=71, reads_seed=72, n_contigs=5,
         min_len=12000, max_len=20000, short_contig_at=None, depth=10,
         notes="fq2 holds 300 records more than fq1 and shorter headers: surplus mate-2 reads are counted in phase A, never voted; "
               "fq1 ends with a blank line"),
    # what the reference ACCEPTS of unequal files (E:356-402)
    Case("k24_fq2_stray2", fq2_stray_records=2, sample=0.5, ref_seed=91, reads_seed=92, n_contigs=5, min_len=12000, max_len=20000,
         short_contig_at=None, depth=14,
         notes="two foreign records in front of fq2: first read IDs differ, phase C re-scans fq2 for fq1's (E:368-402) and pairs fq1's "
               "line g with fq2's line g + 8; phase A counts the foreign records too and samples fq2 by its own ordinals"),
    Case("k24_fq2_short", fq2_drop_tail=500, ref_seed=93, reads_seed=94, n_contigs=5, min_len=12000, max_len=20000, short_contig_at=None,
         depth=10, notes="fq2 ends 500 records early with a newline: the rest of fq1 is voted against an empty mate 2 (E:356-367)"),
    Case("k24_fq2_short_nonl", fq2_drop_tail=500, fq2_last_line_bases_of=21, ref_seed=93, reads_seed=94, n_contigs=5, min_len=12000,
         max_len=20000, short_contig_at=None, depth=10,
         notes="the same with no newline after fq2's last line, which holds bases: std::getline leaves that line in place, so the "
               "rest of fq1 is voted against it"),
    Case("k24_long_line_fq2", sample=0.5, long_line=(2, 40, 620), ref_seed=93, reads_seed=94, n_contigs=5, min_len=12000, max_len=20000,
         short_contig_at=None, depth=10,
         notes="read 40 of fq2 is 620 characters long -- more than the reference's 500-entry buffers, which it fills for sampled reads "
               "only (E:1004-1005, 1044): the read is not sampled, the run is defined (a sampled one ends in a stack overrun)"),
    Case("k24_long_line_fq1", sample=0.5, long_line=(1, 101, 620), ref_seed=93, reads_seed=94, n_contigs=5, min_len=12000, max_len=20000,
         short_contig_at=None, depth=10, notes="the same for read 101 of fq1"),
    # -t N: goldens from the reference with its threads run in creation order (oracle/seq_threads.c)
    Case("k24_t4", threads=4, notes="-t 4: thread chunks of the FASTQs, 2 contig groups, 4 sentinel lines"),
    Case("k24_t8_sample_half", threads=8, sample=0.5, notes="-t 8 with sampling: ordinals restart in every chunk (E:1037)"),
    Case("k24_t3_fq2_longer", threads=3, fq2_header_pad=20, notes="-t 3, fq2 laid out differently: phase C re-synchronises on read IDs (E:368-402), phase A cuts fq2 by fq1's size"),
    Case("k24_t10_sample_bases", threads=10, sample=700000.0, ref_seed=81, reads_seed=82, notes="-t 10 (the CLI default) with --sample > 1"),
    Case("k24_seed7", seed=7, ref_seed=61, reads_seed=62, n_contigs=5, min_len=12000, max_len=20000,
         short_contig_at=None, depth=10, notes="tiny; inputs also committed gz-compressed"),
]}


def sha256_file(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def materialise(case: Case, outdir: str):
    """Write ref.fa / s.1.fq / s.2.fq for the case into outdir; returns the three paths."""
    ref = synth.make_reference(case.ref_seed, case.n_contigs, case.min_len, case.max_len,
                               short_contig_at=case.short_contig_at, n_run_at=case.n_run_at)
    reads = synth.make_sample(ref, case.reads_seed, depth=case.depth, snp_rate=case.snp_rate,
                              n_read_frac=case.n_read_frac)
    return synth.write_case(outdir, ref, reads, fq2_header_pad=case.fq2_header_pad,
                            lowercase_every=case.lowercase_every, fq1_header_pad=case.fq1_header_pad,
                            fq1_drop_tail=case.fq1_drop_tail, fq1_trailing_blank=case.fq1_trailing_blank,
                            fq2_stray_records=case.fq2_stray_records, fq2_drop_tail=case.fq2_drop_tail,
                            fq2_last_line_bases_of=case.fq2_last_line_bases_of, long_line=case.long_line)


def extract_ref_argv(case: Case, fq1: str, fq2: str, fa: str, interval: str):
    """argv[1:] of extract_ref for the case, threads fixed at 1 (parity contract = -t 1)."""
    def num(x):
        return str(int(x)) if float(x) == int(x) else repr(float(x))
    return [fq1, fq2, fa, interval, repr(case.hit_ratio), repr(case.match_ratio), str(case.threads), str(case.k),
            str(case.max_peak), str(case.e), str(case.seed), num(case.sample)]


GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------------- FASTA files with unusual line structure
def odd_fastas(seed: int = 7):
    """[(name, bytes)]: FASTA texts whose line structure exercises read_ref's std::getline semantics (E:761-880): lines before
    the first header (the sequence called "start"), blank lines, CRLF, no final newline, a header as the last line, '>' inside a
    sequence line, adjacent headers, lines as wide as the loader's 4096-byte text blocks, lower case, N runs, headers with the
    three id delimiters of get_read_ID, and two random mixtures.  Bytes stay below 128 (the reference indexes tables by char)."""
    import numpy as np
    rng = np.random.default_rng(seed)

    def seq(n, alphabet=b"ACGT"):
        return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n).tolist())

    def wrap(s, w, eol=b"\n"):
        return b"".join(s[i:i + w] + eol for i in range(0, len(s), w))

    out = []
    out.append(("pre_header", wrap(seq(300), 60) + b">c1 first contig\n" + wrap(seq(1000), 60) + b">c2\tx\n" + wrap(seq(500), 70)))
    out.append(("blank_and_crlf", b">a/1\r\n" + wrap(seq(400), 50, b"\r\n") + b"\r\n\n" + wrap(seq(200), 50) + b"\n\n>b\n" + wrap(seq(300), 61) + b"\n"))
    out.append(("no_final_newline", b">x\n" + wrap(seq(700), 80) + b">y\n" + seq(333)))
    out.append(("header_last", b">x\n" + wrap(seq(500), 80) + b">dangling"))
    out.append(("gt_inside", b">x\n" + seq(100) + b">" + seq(100) + b"\n" + seq(150) + b">\n>y desc>more\n" + wrap(seq(400), 37)))
    out.append(("adjacent_headers", b">a\n>b\n" + wrap(seq(300), 60) + b">c\n\n>d\n" + seq(20) + b"\n>e\n" + wrap(seq(600), 60)))
    long_seq = seq(40000, b"ACGTacgtN")
    wide = b"".join(long_seq[i:j] + b"\n" for i, j in zip([0, 4095, 8191, 12288, 16385, 16386, 24000], [4095, 8191, 12288, 16385, 16386, 24000, 40000]))
    out.append(("wide_lines", b">w\n" + wide + b">n\n" + wrap(b"N" * 100 + seq(900) + b"n" * 40 + seq(300), 64)))
    for r in range(2):
        parts = []
        if r == 0:
            parts.append(wrap(seq(int(rng.integers(1, 200))), 33))
        for c in range(int(rng.integers(5, 40))):
            parts.append(b">r%d_%d%s\n" % (r, c, [b"", b" d", b"/2", b"\tq"][int(rng.integers(0, 4))]))
            n = int(rng.choice([0, 5, 30, 200, 3000, 9000]))
            s = seq(n, b"ACGTACGTACGTacgtN")
            i = 0
            while i < len(s):
                w = int(rng.choice([1, 7, 60, 61, 200, 4096]))
                parts.append(s[i:i + w] + [b"\n", b"\n", b"\r\n", b"\n\n"][int(rng.integers(0, 4))])
                i += w
        text = b"".join(parts)
        out.append((f"random{r}", text if r == 0 else text.rstrip(b"\n")))
    return out
