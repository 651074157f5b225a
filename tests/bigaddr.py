"""The CPU restatement against the HIP path at the ADDRESSES of BASELINE configs[2] / configs[4].

The full-size tests compare the GPU's forms with each other; whatever all forms share above 4 G positions -- ContigDev.hash_word + j*e
into a 156 GB index, flat_base and plane-word arithmetic, 2-D tile numbering -- no independent implementation had looked at.  Phase B
is per contig given the count table (read_index E:888-979, slide_window E:550-725), so the oracle can scan a FEW contigs of the big
reference -- the ones whose flat positions, index words and index bytes straddle 2^31 / 2^32 / 2^33 ... and the last ones -- with
the table exported from the GPU, and every per-position flag, every peak and every registered k-mer of those contigs must be the
GPU's.  Votes are linear in the pairs, so a 200 k-pair subset re-voted by the oracle against the GPU's exported registry checks
phase C at 2^32-slot size."""
import os

import numpy as np


def boundary_contigs(nc, cl, k, e, packed):
    """0-based contigs whose addresses cross a power of two in some unit a kernel computes in, plus the first and the last ones"""
    words_per_contig = 1 + (cl - k + 1) * e
    picks = {0, 1, nc - 2, nc - 1}
    for p in range(30, 40):
        for unit in ((1,) if packed else (1, words_per_contig / cl, 4 * words_per_contig / cl)):   # flat positions, index words, index bytes
            c = int((1 << p) / (unit * cl))
            if 0 < c < nc - 1:
                picks.update((c - 1, c, c + 1) if p == 32 else (c,))
    if packed:          # word index of plane m at position x: m * plane_words + (x >> 5): where that crosses 2^31 / 2^32
        pw = (nc * cl + 31) // 32 + 2
        for m in (1, 2):
            for lim in (1 << 31, 1 << 32):
                x = (lim - m * pw) * 32
                if 0 < x < nc * cl:
                    picks.add(int(x // cl))
    return sorted(picks)


def check_against_oracle(eng, oracle, tmp, nc, cl, k, e, contigs, vote_pairs=0, sample_contigs=300):
    """`eng` holds the whole reference and a counted sample.  Returns (n_positions_checked, n_peaks_checked, n_votes_checked)."""
    from localhgt_amd.engine import Engine
    import bench
    cc = eng.coder_get()
    table = eng.counts_export()                                  # the GPU's 2^k-slot table as the oracle's u8 array
    fa = os.path.join(tmp, f"subset_k{k}.fa")
    with Engine(k, e, device=eng.device) as e2, open(fa, "wb") as f:
        e2.rng_seed(1)
        e2.coder_generate()
        assert (e2.coder_get() == cc).all()
        for c in contigs:                                         # contig c alone = shard c of nc
            bases = e2.synth_reference_shard(1, nc, cl, c, nc, want_host=True)
            assert bases.size == cl
            f.write(b">c%d\n" % c)
            f.write(bases.tobytes())
            f.write(b"\n")
    idx = fa + ".index.dat"
    assert oracle.index_build(fa, idx, fa + ".genome.len.txt", k, e, cc) == len(contigs)
    n_sub = len(contigs) * cl
    flags_o = np.zeros(n_sub, dtype=np.uint8)
    pk_o = np.zeros(1 << k, dtype=np.uint32)
    n_o, loci_o, _ = oracle.ref_scan(idx, table, k, e, np.float32(0.1), np.float32(0.08), 3_000_000, pk_o, flags_o)
    del table
    inside_o = (flags_o >> 2) & 1
    checked = 0
    for form, dbg in (("exact", 8192), ("picked", 0)):
        eng.set_debug(dbg)
        n_g = eng.ref_scan(0.1, 0.08, 300_000_000)
        eng.set_debug(0)
        for i, c in enumerate(contigs):
            fg = eng.flags_export(c * cl, cl)
            fo, io = flags_o[i * cl:(i + 1) * cl], inside_o[i * cl:(i + 1) * cl]
            if form == "exact":
                assert ((fg & 0b11) == (fo & 0b11)).all(), (form, c, "single/trio", int(((fg & 3) != (fo & 3)).sum()))
            else:         # the form the engine picks: `single` is exact where bit 7 says so (single-first: everywhere), a lower bound elsewhere
                ex = (fg & 0x80) != 0
                assert ((fg & 1)[ex] == (fo & 1)[ex]).all() and ((fg & 1) <= (fo & 1)).all(), (form, c, "single")
                assert ((fg & 2)[ex] == (fo & 2)[ex]).all() or eng.scan_info()["form"] == "single-first", (form, c, "trio")
            assert (((fg >> 4) & 1) == io).all(), (form, c, "inside a good interval", int((((fg >> 4) & 1) != io).sum()))
            assert (((fg >> 3) & 1) == (((fo >> 3) & 1) & io)).all(), (form, c, "peak")
            checked += cl
        loci_g, _ = eng.peaks_export(n_g)
        cg, pg = loci_g[0::2], loci_g[1::2]
        co, po = loci_o[0:2 * n_o:2], loci_o[1:2 * n_o:2]
        for i, c in enumerate(contigs):                           # the oracle numbers the subset's contigs 1, 2, ...; the GPU all of them
            assert (pg[cg == c + 1] == po[co == i + 1]).all() and (cg == c + 1).sum() == (co == i + 1).sum(), (form, c, "peak positions")
    assert n_o > 0 and inside_o.any(), "the chosen contigs show nothing: the sample does not reach them"
    # registry: every slot the subset's peaks registered holds, on the GPU, an id at least that of the same peak (a later contig's
    # peak may have overwritten it: larger id wins, E:262) -- the whole 2^k table comes over
    pk_g = eng.peak_kmer_export()
    ids_g = np.flatnonzero(np.isin(cg, np.asarray(contigs) + 1))   # GPU ids of the subset's peaks, ascending like the oracle's 0 .. n_o-1
    assert ids_g.size == n_o
    slots = np.flatnonzero(pk_o)
    assert (pk_g[slots] >= ids_g[pk_o[slots]]).all() and (pk_g[slots] != 0).all()
    last = np.flatnonzero(co == len(contigs))                      # peaks of the LAST contig of the reference: nothing overwrites them
    if contigs[-1] == nc - 1 and last.size:
        s_last = slots[np.isin(pk_o[slots], last)]
        assert (pk_g[s_last] == ids_g[pk_o[s_last]]).all()
    n_votes = 0
    if vote_pairs:
        # phase C at full table size: the oracle re-votes a subset of the pairs against the GPU's exported registry
        first = 1_234_567
        eng.pairs_clear()
        eng.synth_options(0, 20, sample_contigs)
        m1, m2 = eng.synth_pairs(1, 2, nc, cl, first, vote_pairs, 150, want_host=True)
        eng.synth_options(0, 20, 0)
        f1, f2 = os.path.join(tmp, "sub.1.fq"), os.path.join(tmp, "sub.2.fq")
        bench.write_fastq(f1, m1, vote_pairs, 150, "1")
        bench.write_fastq(f2, m2, vote_pairs, 150, "2")
        eng.vote()
        _, pf_g = eng.peaks_export(n_g)
        kept, pf_o = oracle.vote(f1, f2, k, e, cc, 100.0, None, pk_g, loci_g, n_g, threads=os.cpu_count() or 1)
        assert kept == vote_pairs
        assert (pf_g == pf_o[:n_g]).all(), int((pf_g != pf_o[:n_g]).sum())
        n_votes = int((pf_g > 0).sum())
        assert n_votes > 0, "the subset of pairs votes for nothing: the check would be empty"
    return checked, int(n_o), n_votes
