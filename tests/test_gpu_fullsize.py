"""Parity at BASELINE.json's full single-GPU size (1 Gbase reference, 10 M pairs, k=32) through
size-independent properties: the oracle cannot run here in seconds, so the two independent GPU count
paths are compared with each other (the direct kernel is oracle-checked at small sizes), and
sharding / linearity identities are checked on the real kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NC, CL, NP, K, E = 1000, 1_000_000, 10_000_000, 32, 3


@pytest.fixture(scope="module")
def eng():
    from localhgt_amd.engine import Engine
    e = Engine(K, E)
    e.rng_seed(1)
    e.coder_generate()
    e.synth_reference(1, NC, CL)
    yield e
    e.close()


def _slices(eng):
    """histogram + three 16 M-slot windows of the table (start, middle, end)"""
    parts = [eng.counts_histogram().astype(np.uint64)]
    for first in (0, (1 << 31) - (1 << 23), (1 << 32) - (1 << 24)):
        parts.append(eng.counts_export(first, 1 << 24).astype(np.uint64))
    return np.concatenate(parts)


def test_partitioned_equals_direct_at_full_size(eng):
    eng.pairs_clear()
    eng.synth_pairs(1, 2, NC, CL, 0, NP)          # 3 partition chunks of <= 4 M pairs
    got = []
    for mode in (1, 0):
        eng.set_count_mode(mode)
        eng.counts_clear()
        eng.count_kmers()
        got.append(_slices(eng))
    eng.set_count_mode(-1)
    assert (got[0] == got[1]).all()
    assert got[0][:4].sum() == 1 << 32 and got[0][3] > 0


def test_shards_merge_to_whole_and_votes_are_linear(eng):
    """count(A) (+) count(B) == count(A u B) with saturating merge; votes(A u B) == votes(A) + votes(B)"""
    from localhgt_amd.engine import Engine
    half = NP // 2
    eng.pairs_clear()
    eng.synth_pairs(1, 2, NC, CL, 0, NP)
    eng.counts_clear()
    eng.count_kmers()
    whole = _slices(eng)
    n_peaks = eng.ref_scan(0.1, 0.08, 300_000_000)
    eng.vote()
    loci_w, votes_w = eng.peaks_export(n_peaks)
    assert n_peaks > 1000 and votes_w.max() >= 1
    # shard B on a second engine sharing nothing but the coder
    with Engine(K, E) as eb:
        eb.coder_set(eng.coder_get())
        eb.synth_pairs(1, 2, NC, CL, half, NP - half)
        eb.count_kmers()
        eng.pairs_clear()
        eng.synth_pairs(1, 2, NC, CL, 0, half)
        eng.counts_clear()
        eng.count_kmers()
        pb, nb = eb.counts_buffer()
        eng.counts_merge(pb, 0, nb)
    assert (_slices(eng) == whole).all()
    assert eng.ref_scan(0.1, 0.08, 300_000_000) == n_peaks          # same table -> same peaks
    loci_a, _ = eng.peaks_export(n_peaks)
    assert (loci_a == loci_w).all()
    eng.vote()
    va = eng.peaks_export(n_peaks)[1].astype(int)
    eng.pairs_clear()
    eng.synth_pairs(1, 2, NC, CL, half, NP - half)
    eng.ref_scan(0.1, 0.08, 300_000_000)                             # clears the vote counters
    eng.vote()
    vb = eng.peaks_export(n_peaks)[1].astype(int)
    assert (np.minimum(254, va + vb) == votes_w).all()
    # sortedness / structure of the registry: ids ascend in (contig, position), one new peak per 50-bp bucket
    contig, pos = loci_w[0::2].astype(np.int64), loci_w[1::2].astype(np.int64)
    key = contig * (1 << 32) + pos
    assert (np.diff(key) > 0).all()
    same = np.diff(contig) == 0
    assert (np.diff(pos // 50)[same] > 0).all()


def test_settled_tiles_change_nothing_on_a_saturated_table(eng):
    """every k-mer of the sampled half of the reference at count 3: its 2000-position tiles are settled by window_good alone (all
    positions good, contrast neighbourhood fully hit) and never reach interval_select; the other half is hit by collisions only.
    Same flags for every position and same peaks as with that shortcut switched off"""
    eng.counts_clear()
    for part in range(4):                        # 40 M different pairs (24x coverage), each batch counted three times: hit slots end at 3
        eng.pairs_clear()
        eng.synth_pairs(1, 2, NC, CL, part * NP, NP)
        for _ in range(3):
            eng.count_kmers()
    hist = eng.counts_histogram()
    assert hist[3] > 0.2 * (1 << 32) and hist[1] == hist[2] == 0, hist
    res = []
    for flags in (8192, 8192 | 256, 4096, 16384, 1 << 24, 0):    # exact scan with / without settled tiles, single-first, trio-first, the same from the slot list, the form the engine picks
        eng.set_debug(flags)
        n = eng.ref_scan(0.1, 0.08, 300_000_000)
        res.append((n, eng.peaks_export(n)[0].copy(), eng.flags_export(0, NC * CL) & 0b1111100))   # single / trio are bounds outside the tiles a lite form treats exactly
    info = eng.scan_info()
    assert 0.2 < info["frac_slots_at_3"] < 0.9 and info["tiles"] == NC * CL // 2000   # in the range where the trial decides
    eng.set_debug(0)
    for other in res[1:]:
        assert res[0][0] == other[0]
        assert (res[0][1] == other[1]).all()
        assert (res[0][2] == other[2]).all()
    inside = (res[0][2] >> 4) & 1
    assert 0.3 < inside.mean() < 0.7             # the sampled contigs lie inside good intervals, the others do not


def test_count_on_load_equals_count_after_load(eng, tmp_path):
    """the FASTQ pipeline with phase A running behind the parser (1 Mi-pair batches counted as they become resident) gives the table
    of load-everything-then-count, and lhgt_count_kmers neither counts those batches again nor skips them the next time"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = 2_300_000                                   # three batches with count-on-load, one without
    eng.pairs_clear()
    m1, m2 = eng.synth_pairs(1, 7, NC, CL, 0, n, 150, want_host=True)
    eng.pairs_clear()
    f1, f2 = str(tmp_path / "c.1.fq"), str(tmp_path / "c.2.fq")
    bench.write_fastq(f1, m1, n, 150, "1")
    bench.write_fastq(f2, m2, n, 150, "2")
    del m1, m2
    got = []
    for on in (True, False):
        eng.counts_clear()
        eng.set_count_on_load(on)
        seen, kept = eng.pairs_load_fastq(f1, f2, 100.0)
        assert seen == kept == n == eng.pairs_count()
        eng.count_kmers()
        got.append((eng.digest(eng.DIGEST_COUNTS), tuple(int(x) for x in eng.counts_histogram())))
        if on:                                      # a second count of the same resident pairs really counts them again
            before = eng.counts_histogram()
            eng.count_kmers()
            after = eng.counts_histogram()
            assert after[3] > before[3] and after[0] == before[0]
        eng.pairs_clear()
    eng.set_count_on_load(False)
    assert got[0] == got[1]
    assert got[0][1][0] < (1 << 32) and got[0][1][1] > 0


def test_deep_sample_on_a_dense_peak_set_shares_its_line_fills(eng):
    """round 6: 10 M pairs from 30 genomes of the 1 Gbase reference (100x) with SNPs at 1 % of the sample genomes' bases -- the
    reference's own read model (test/run_BKP_detection.sh: species20_snp0.01) -- against a peak set voted without a bitmap (debug
    bit 2).  The engine picks the shared-line-fill form by itself (40 reads per champion k-mer), mixes two batches (a 250-base pair
    in each goes to the generic kernel behind it), and the votes are those of the dense generic kernel (bit 28: never shared)."""
    eng.pairs_clear()
    eng.synth_options(10, 20, 30)
    eng.synth_read_mix(1, 250)                    # one pair in a thousand with 250-base reads: the form's fallback list
    eng.synth_pairs(1, 2, NC, CL, 0, 6_000_000)
    eng.synth_pairs(1, 3, NC, CL, 6_000_000, 4_000_000)
    eng.synth_read_mix(0, 0)
    eng.synth_options(0, 20, 0)
    eng.counts_clear()
    eng.count_kmers()
    got = {}
    for name, dbg in (("shared", 4), ("dense", 4 | (1 << 28)), ("picked", 0)):
        eng.set_debug(dbg)
        n = eng.ref_scan(0.1, 0.08, 300_000_000)
        eng.work_stats(1)
        eng.vote()
        st = eng.work_stats(0)
        got[name] = (n, eng.digest(eng.DIGEST_VOTES), eng.vote_info()["form"], st["vote_shared_fetches"])
        eng.set_debug(0)
    assert got["shared"][2] == "shared" and got["dense"][2] == "dense", got
    assert got["shared"][:2] == got["dense"][:2] == got["picked"][:2], got
    assert got["shared"][1][1] >= 10, "nothing voted: the check would be empty"
    probes = 10_000_000 * 714
    assert 0 < got["shared"][3] < probes / 4, f"{got['shared'][3]} line fills for {probes} probes: the reads do not share"
