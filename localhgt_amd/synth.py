"""Synthetic references and paired reads for tests and bench.py.

The reference repository ships no data for its smoke test (test/ref.fa and both FASTQs
are absent), so every input used here is generated from a seed:

* reference: contigs of iid uniform ACGT, 60-column FASTA;
* sample: the first half of the contigs, taken in (recipient, donor) pairs; a 3 kb
  segment is cut out of the donor and pasted into the recipient, so both junction
  kinds (insertion and deletion) show up as coverage edges; fragments of 300-500 bp,
  150 bp mates, mate 2 reverse-complemented, constant qualities, optional SNPs and a
  fraction of reads carrying one `N` (mirrors paper_results/simulation.py:280-299).

Everything is numpy (PCG64 `default_rng`) and deterministic for a given seed.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


@dataclass
class SynthRef:
    names: List[str]
    seqs: List[np.ndarray]  # each uint8 codes 0..3 (A,C,G,T); value 4 = 'N'

    @property
    def total_len(self) -> int:
        return int(sum(len(s) for s in self.seqs))


def make_reference(seed: int, n_contigs: int, min_len: int, max_len: int,
                   short_contig_at: Optional[int] = None, short_len: int = 20,
                   n_run_at: Optional[Tuple[int, int, int]] = None) -> SynthRef:
    """`short_contig_at` inserts one contig of `short_len` (< k) bases at that list position
    (quirk Q7); `n_run_at=(contig, pos, length)` overwrites a run with N (quirk Q6)."""
    rng = np.random.default_rng(seed)
    names, seqs = [], []
    for i in range(n_contigs):
        ln = int(rng.integers(min_len, max_len + 1))
        seqs.append(rng.integers(0, 4, size=ln, dtype=np.uint8))
        names.append(f"g{i + 1}")
    if n_run_at is not None:
        c, p, ln = n_run_at
        seqs[c][p:p + ln] = 4
    if short_contig_at is not None:
        seqs.insert(short_contig_at, rng.integers(0, 4, size=short_len, dtype=np.uint8))
        names.insert(short_contig_at, "gshort")
    return SynthRef(names, seqs)


def codes_to_ascii(codes: np.ndarray) -> np.ndarray:
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    return lut[codes]


def write_fasta(ref: SynthRef, path: str, width: int = 60, name_suffix: str = "") -> None:
    with open(path, "wb") as f:
        for name, seq in zip(ref.names, ref.seqs):
            f.write(b">" + name.encode() + name_suffix.encode() + b"\n")
            a = codes_to_ascii(seq)
            n_full = len(a) // width
            if n_full:
                body = np.empty((n_full, width + 1), dtype=np.uint8)
                body[:, :width] = a[: n_full * width].reshape(n_full, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if len(a) % width:
                f.write(a[n_full * width:].tobytes() + b"\n")


@dataclass
class SynthReads:
    mate1: np.ndarray  # [n, L] uint8 codes (4 = N)
    mate2: np.ndarray
    truth: List[Tuple[str, int]] = field(default_factory=list)  # (contig name, breakpoint pos)


def _revcomp(codes: np.ndarray) -> np.ndarray:
    out = codes[..., ::-1].copy()
    valid = out < 4
    out[valid] = 3 - out[valid]
    return out


def make_sample(ref: SynthRef, seed: int, depth: float, read_len: int = 150,
                frag_min: int = 300, frag_max: int = 500, transfer_len: int = 3000,
                snp_rate: float = 0.0, n_read_frac: float = 0.02,
                n_pairs: Optional[int] = None) -> SynthReads:
    """Reads from the first half of the (long enough) contigs with cut-and-paste transfers."""
    rng = np.random.default_rng(seed)
    usable = [i for i, s in enumerate(ref.seqs) if len(s) > 4 * transfer_len]
    chosen = usable[: max(2, (len(usable) // 2) // 2 * 2)]
    genomes, truth = [], []
    for a in range(0, len(chosen) - 1, 2):
        ri, di = chosen[a], chosen[a + 1]
        rec, don = ref.seqs[ri], ref.seqs[di]
        d0 = int(rng.integers(transfer_len, len(don) - 2 * transfer_len))
        r0 = int(rng.integers(transfer_len, len(rec) - transfer_len))
        seg = don[d0:d0 + transfer_len]
        if rng.random() < 0.5:
            seg = _revcomp(seg)
        genomes.append(np.concatenate([rec[:r0], seg, rec[r0:]]))
        genomes.append(np.concatenate([don[:d0], don[d0 + transfer_len:]]))
        truth += [(ref.names[ri], r0), (ref.names[di], d0), (ref.names[di], d0 + transfer_len)]
    if len(chosen) % 2:
        genomes.append(ref.seqs[chosen[-1]])
    lens = np.array([len(g) for g in genomes], dtype=np.int64)
    if n_pairs is None:
        n_pairs = int(depth * lens.sum() / (2 * read_len))
    # fragments: genome chosen proportionally to length
    gsel = rng.choice(len(genomes), size=n_pairs, p=lens / lens.sum())
    flen = rng.integers(frag_min, frag_max + 1, size=n_pairs)
    u = rng.random(n_pairs)
    flip = rng.random(n_pairs) < 0.5
    m1 = np.empty((n_pairs, read_len), dtype=np.uint8)
    m2 = np.empty((n_pairs, read_len), dtype=np.uint8)
    ar = np.arange(read_len)
    for gi, g in enumerate(genomes):
        sel = np.nonzero(gsel == gi)[0]
        if sel.size == 0:
            continue
        fl = np.minimum(flen[sel], len(g))
        st = (u[sel] * (len(g) - fl + 1)).astype(np.int64)
        left = g[st[:, None] + ar[None, :]]
        right = _revcomp(g[(st + fl - read_len)[:, None] + ar[None, :]])
        f = flip[sel]
        m1[sel] = np.where(f[:, None], right, left)
        m2[sel] = np.where(f[:, None], left, right)
    if snp_rate > 0:
        for m in (m1, m2):
            hit = rng.random(m.shape) < snp_rate
            m[hit] = (m[hit] + rng.integers(1, 4, size=int(hit.sum()), dtype=np.uint8)) % 4
    if n_read_frac > 0:
        for m in (m1, m2):
            rows = np.nonzero(rng.random(n_pairs) < n_read_frac)[0]
            cols = rng.integers(0, read_len, size=rows.size)
            m[rows, cols] = 4
    return SynthReads(m1, m2, truth)


def write_fastq(mate: np.ndarray, path: str, suffix: str, header_pad: int = 0,
                lowercase_every: int = 0) -> None:
    """4-line records `@r<9-digit ordinal>/<suffix>`; `header_pad` appends that many bytes of
    comment to every header (used to make fq2 longer than fq1, quirk Q4)."""
    n, L = mate.shape
    a = codes_to_ascii(mate)
    if lowercase_every:
        a = a.copy()
        a[::lowercase_every] |= 0x20
    ids = np.char.zfill(np.arange(n).astype("U9"), 9)
    head = np.char.add(np.char.add("@r", ids), "/" + suffix + (" " + "x" * (header_pad - 1) if header_pad else ""))
    hb = np.frombuffer("".join(head.tolist()).encode(), dtype=np.uint8).reshape(n, -1)
    hl = hb.shape[1]
    rec = np.empty((n, hl + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, :hl] = hb
    rec[:, hl] = 10
    rec[:, hl + 1: hl + 1 + L] = a
    rec[:, hl + 1 + L] = 10
    rec[:, hl + 2 + L] = ord("+")
    rec[:, hl + 3 + L] = 10
    rec[:, hl + 4 + L: hl + 4 + 2 * L] = ord("I")
    rec[:, hl + 4 + 2 * L] = 10
    with open(path, "wb") as f:
        f.write(rec.tobytes())


def write_case(outdir: str, ref: SynthRef, reads: SynthReads, fq2_header_pad: int = 0,
               lowercase_every: int = 0, fq1_header_pad: int = 0, fq1_drop_tail: int = 0,
               fq1_trailing_blank: bool = False, fq2_stray_records: int = 0, fq2_drop_tail: int = 0,
               fq2_last_line_bases_of=None, long_line=None) -> Tuple[str, str, str]:
    """fq1_drop_tail: fq1 loses its last records, so fq2 holds surplus ones (counted in phase A while they start inside
    size(fq1), E:1438-1445; never voted, E:356); fq1_trailing_blank: one empty line after fq1's last record;
    fq2_stray_records: that many records with foreign read IDs in front of fq2 (copies of its last reads), so the first IDs differ
    and phase C re-scans fq2 for fq1's (E:368-402); fq2_drop_tail: fq2 loses its last records, so phase C runs out of mate-2 lines
    (E:356-367); fq2_last_line_bases_of = r: fq2's last line (a quality line) holds the bases of mate 2 of read r and has no newline --
    the line std::getline then leaves behind for every later read of fq1; long_line = (file 1 or 2, read r, length): that read's
    sequence and quality lines in that file are `length` characters long -- more than the reference's 500-entry buffers, which it
    fills for sampled reads only (E:1004-1005, 1044)"""
    os.makedirs(outdir, exist_ok=True)
    fa = os.path.join(outdir, "ref.fa")
    f1 = os.path.join(outdir, "s.1.fq")
    f2 = os.path.join(outdir, "s.2.fq")
    write_fasta(ref, fa)
    m1 = reads.mate1[:len(reads.mate1) - fq1_drop_tail] if fq1_drop_tail else reads.mate1
    write_fastq(m1, f1, "1", header_pad=fq1_header_pad, lowercase_every=lowercase_every)
    if fq1_trailing_blank:
        with open(f1, "ab") as f:
            f.write(b"\n")
    m2 = reads.mate2[:len(reads.mate2) - fq2_drop_tail] if fq2_drop_tail else reads.mate2
    write_fastq(m2, f2, "2", header_pad=fq2_header_pad)
    if fq2_stray_records or fq2_last_line_bases_of is not None:
        body = open(f2, "rb").read()
        if fq2_last_line_bases_of is not None:
            lines = body.split(b"\n")          # ..., header, bases, +, quality, ""
            lines[-2] = codes_to_ascii(reads.mate2[fq2_last_line_bases_of:fq2_last_line_bases_of + 1])[0].tobytes()
            body = b"\n".join(lines[:-1])
        if fq2_stray_records:
            a = codes_to_ascii(reads.mate2[-fq2_stray_records:])
            stray = b"".join(b"@stray%06d/2\n" % i + a[i].tobytes() + b"\n+\n" + b"I" * a.shape[1] + b"\n" for i in range(fq2_stray_records))
            body = stray + body
        with open(f2, "wb") as f:
            f.write(body)
    if long_line is not None:
        which, r, length = long_line
        path = f1 if which == 1 else f2
        lines = open(path, "rb").read().split(b"\n")
        seq = lines[4 * r + 1]
        lines[4 * r + 1] = (seq * (length // max(1, len(seq)) + 1))[:length]
        lines[4 * r + 3] = b"I" * length
        with open(path, "wb") as f:
            f.write(b"\n".join(lines))
    return fa, f1, f2


def ragged_cuts(total: int, seed: int = 5) -> np.ndarray:
    """contig boundaries of a catalogue-like length distribution over `total` bases (for lhgt_synth_reference_cuts): 3 of 4
    contigs log-uniform in 300 b .. 20 kb, the others 20 kb .. 2 Mb, and one piece in a hundred of 10 .. 32 bases (not indexed at
    k = 32, E:772) -- what a real catalogue (UHGG: hundreds of thousands of contigs) looks like next to 13000 x 1 Mbp"""
    rng = np.random.default_rng(seed)
    lens = []
    have = 0
    while have < total:
        n = 4096
        u = rng.random(n)
        kind = rng.random(n)
        ln = np.where(kind < 0.75, np.exp(np.log(300) + u * np.log(20000 / 300)), np.exp(np.log(20000) + u * np.log(2_000_000 / 20000)))
        ln = np.where(rng.random(n) < 0.01, 10 + (u * 23), ln).astype(np.int64)
        lens.append(ln)
        have += int(ln.sum())
    lens = np.concatenate(lens)
    cuts = np.concatenate([[0], np.cumsum(lens)])
    cuts = cuts[cuts < total]
    return np.concatenate([cuts, [total]]).astype(np.uint64)
