"""BASELINE.json configs[4]: a reference of more than 50 Gbase on ONE GPU, possible with the packed resident form (3/8 byte per
base).  Its own module: the 13 Gbase engine of test_gpu_fullsize_uhgg.py (156 GB of index) must be gone before this one
allocates its per-position arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CL, E = 1_000_000, 3
FLAG_BITS = 0b1111100


def _scan(eng, debug):
    eng.set_debug(debug)
    n = eng.ref_scan(0.1, 0.08, 300_000_000)
    info = eng.scan_info()
    res = (n, eng.digest(eng.DIGEST_LOCI), eng.digest(eng.DIGEST_PEAK_KMER), eng.digest(eng.DIGEST_FLAGS, FLAG_BITS))
    eng.set_debug(0)
    return res, info


def _vote(eng, debug):
    eng.set_debug(debug)
    eng.ref_scan(0.1, 0.08, 300_000_000)
    eng.vote()
    d = eng.digest(eng.DIGEST_VOTES)
    eng.set_debug(0)
    return d


@pytest.mark.parametrize("k", [32, 21])
def test_progenomes_scale_reference_on_one_gpu(k, oracle, tmp_path):
    """BASELINE configs[4] names a reference of more than 50 GB, whose index (12 bytes per base: 600 GB) only fits sharded over
    eight GPUs.  Packed, 50 Gbase are 19 GB: the whole reference, its per-position flags and the tables fit ONE GPU.  Forms of
    the scan and of the vote against each other, as at 13 Gbase, for both ends of the config's k = 21 / 32 sweep (at k = 21 the
    2^21-slot table is full after the first reads: every window is good, the contrast is flat, peaks only at contig ends).
    The sample: 10 M pairs from 300 of the genomes -- a denser one turns every window of 50 Gbase good and overflows max_peak."""
    from localhgt_amd.engine import Engine
    nc = 50_000
    with Engine(k, E) as e:
        e.rng_seed(1)
        e.coder_generate()
        e.set_reference_form(True)
        e.synth_reference(1, nc, CL)
        info = e.reference_info()
        assert info["form"] == "packed" and 18e9 < info["resident_bytes"] < 20e9
        e.synth_options(0, 20, 300)
        e.synth_pairs(1, 2, nc, CL, 0, 10_000_000)
        e.synth_options(0, 20, 0)
        e.count_kmers()
        exact, info_x = _scan(e, 8192)
        assert info_x["tiles"] == nc * CL // 2000 and exact[0] > 1000, (info_x, exact)
        for dbg in (0, 4096, 16384):
            got, sinfo = _scan(e, dbg)
            assert got == exact, (dbg, sinfo)
        votes = _vote(e, 0)
        assert votes == _vote(e, 32) == _vote(e, 4) == _vote(e, 1 << 19) and (votes[1] >= 1 or k < 32), votes   # bit 19: every pair walked, no bound
        # configs[4] AS NAMED (round 6): 200 M input pairs under the CLI's default --sample 2000000000 -- the reference's own answer to a
        # sample too dense for max_peak (E:1392-1398, scripts/infer_HGT_breakpoint.py:209): ratio = 2e9 / (2 x 200 M x 150) = 3.33 %, i.e.
        # 6 666 666 pairs pass the sampling array whatever the input size; a subset of iid pairs is iid, so the kept pairs are generated
        # directly.  The same properties on that sample: forms of the scan and of the vote agree, max_peak (the default 3 x 10^8) holds
        kept = int(2e9 / (2 * 150))
        e.pairs_clear()
        e.synth_options(0, 20, 300)
        e.synth_pairs(1, 3, nc, CL, 0, kept)
        e.synth_options(0, 20, 0)
        e.counts_clear()
        e.count_kmers()
        named, _ = _scan(e, 8192)
        assert 0 < named[0] < 300_000_000
        for dbg in (0, 16384):
            got, sinfo = _scan(e, dbg)
            assert got == named, ("as named", dbg, sinfo)
        assert _vote(e, 0) == _vote(e, 4), "as named"
        # (back to the 10 M-pair sample for the address checks below)
        e.pairs_clear()
        e.synth_options(0, 20, 300)
        e.synth_pairs(1, 2, nc, CL, 0, 10_000_000)
        e.synth_options(0, 20, 0)
        e.counts_clear()
        e.count_kmers()
        n_peaks = e.ref_scan(0.1, 0.08, 300_000_000)
        loci, _ = e.peaks_export(n_peaks)
        contig, pos = loci[0::2].astype(np.int64), loci[1::2].astype(np.int64)
        assert (np.diff(contig * (1 << 32) + pos) > 0).all() and contig.max() <= nc and pos.max() < CL
        # the CPU restatement on the contigs whose plane words straddle 2^31 / 2^32 and whose flat positions straddle 2^32 .. 2^35
        # (tests/bigaddr.py): flags, peaks and registry of those contigs; at k = 32 also the votes of 100 000 pairs
        import bigaddr
        contigs = bigaddr.boundary_contigs(nc, CL, k, E, True)
        assert 37438 in contigs and 18719 in contigs and 34359 in contigs
        checked, n_peaks, n_votes = bigaddr.check_against_oracle(e, oracle, str(tmp_path), nc, CL, k, E, contigs, vote_pairs=100_000 if k == 32 else 0)
        assert checked == 2 * len(contigs) * CL and n_peaks >= 1
