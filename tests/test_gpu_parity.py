"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the reference goldens.
Bit-exact everywhere: this path is integer / index work."""
import json
import os
import shutil

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    from localhgt_amd.engine import Engine as E
    return E


# ------------------------------------------------------------------ row H: hashes
@pytest.mark.parametrize("k,e,seed", [(24, 3, 1), (32, 3, 1), (21, 3, 9), (20, 2, 4), (22, 5, 2), (8, 9, 3), (31, 1, 5)])
def test_hash_matches_oracle(Engine, oracle, k, e, seed):
    rng = np.random.default_rng(100 + k + e)
    seq = rng.choice(np.frombuffer(b"ACGTacgtNRn-", dtype=np.uint8), size=700,
                     p=[.22, .22, .22, .22, .02, .02, .02, .02, .01, .01, .01, .01]).tobytes()
    with Engine(k, e) as eng:
        eng.rng_seed(seed)
        eng.coder_generate()
        cc = eng.coder_get()
        got, valid = eng.hash_sequence(seq)
    assert got.shape == (len(seq) - k + 1, e)
    for j in range(len(seq) - k + 1):
        ok, h = oracle.hash_kmer(seq[j:j + k], k, e, cc)
        assert ok == valid[j], j
        if ok:
            assert (got[j] == h).all(), (j, got[j], h)
        else:
            assert (got[j] == 0).all()          # index convention: invalid -> 0 (quirk Q6)


def test_hash_long_contig_word_boundaries(Engine, oracle):
    """every in-word offset, plus a sequence long enough to cross many 32-base words"""
    k, e = 32, 3
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5000)].tobytes()
    with Engine(k, e) as eng:
        eng.rng_seed(3)
        eng.coder_generate()
        cc = eng.coder_get()
        got, valid = eng.hash_sequence(seq)
    assert valid.all()
    for j in list(range(0, 140)) + list(range(4900, 5000 - k + 1)):
        ok, h = oracle.hash_kmer(seq[j:j + k], k, e, cc)
        assert ok and (got[j] == h).all()


# ------------------------------------------------------------------ stepwise parity on one case
@pytest.mark.parametrize("name", ["k24_base", "k24_nrun_lower", "k20_e2"])
def test_phases_match_oracle(Engine, oracle, case_inputs, name, tmp_path):
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    k, e = case.k, case.e
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    index = f"{fa2}.k{k}.h{e}.index.dat"
    with Engine(k, e) as eng:
        # R + I
        eng.rng_seed(case.seed)
        eng.coder_generate()
        oracle.srand(case.seed)
        cc = oracle.random_coder(k, e)
        assert (eng.coder_get() == cc).all()
        n_contigs, n_bases = eng.index_build(fa2, index, fa2 + ".genome.len.txt")
        assert cases.sha256_file(index) == meta["sha256"]["index.dat"]
        assert open(fa2 + ".genome.len.txt").read() == open(os.path.join(cases.GOLDEN_DIR, name, "genome.len.txt")).read()
        assert eng.index_load(index) == (n_contigs, n_bases)
        # A
        ratio = eng.sam_ratio(f1, case.sample)
        assert ratio == oracle.sam_ratio(f1, float(case.sample))
        eng.sampling_init(ratio)
        seen, kept = eng.pairs_load_fastq(f1, f2, ratio)
        eng.count_kmers()
        table = np.zeros(1 << k, dtype=np.uint8)
        size1 = os.path.getsize(f1)
        c1 = oracle.count(f1, size1, k, e, cc, ratio, None, table)
        oracle.count(f2, size1, k, e, cc, ratio, None, table)
        assert kept == c1 == seen
        got = eng.counts_export()
        assert (got == table).all(), f"{int((got != table).sum())} slots differ"
        assert (eng.counts_histogram() == np.bincount(table, minlength=4)).all()
        # B
        flags_o = np.zeros(n_bases, dtype=np.uint8)
        pk_o = np.zeros(1 << k, dtype=np.uint32)
        n_o, loci_o, _ = oracle.ref_scan(index, table, k, e, np.float32(case.hit_ratio), np.float32(case.match_ratio),
                                         case.max_peak, pk_o, flags_o)
        eng.set_debug(8192)                               # the exact form: every hash of every position probed
        n_g = eng.ref_scan(case.hit_ratio, case.match_ratio, case.max_peak)
        assert eng.scan_info()["form"] == "exact"
        eng.set_debug(0)
        flags_g = eng.flags_export(0, n_bases)
        assert ((flags_g & 0b0011) == (flags_o & 0b0011)).all(), "single/trio flags differ"
        inside_o = (flags_o >> 2) & 1
        assert (((flags_g >> 4) & 1) == inside_o).all(), "good-interval mask differs"
        # the contrast peak flag is only computed (and only ever used, E:688-692) inside good intervals
        assert (((flags_g >> 3) & 1) == (((flags_o >> 3) & 1) & inside_o)).all(), "peak flags differ"
        assert n_g == n_o == meta["raw_peaks"]
        loci_g, _ = eng.peaks_export(n_g)
        assert (loci_g == loci_o[:2 * n_o]).all()
        assert (eng.peak_kmer_export() == pk_o).all()
        # C
        eng.vote()
        _, pf_o = oracle.vote(f1, f2, k, e, cc, ratio, None, pk_o, loci_o, n_o)
        _, pf_g = eng.peaks_export(n_g)
        assert (pf_g == pf_o[:n_g]).all()
        # every vote kernel (queued with its direct branch forced, generic with / without the bitmap, no LDS fold); the form of the scan the
        # engine picks by itself (0), the single-first ("lite") form, also with no tile settled early, and the trio-first form
        # ... and (round 6) the shared-line-fill form of a dense peak set (bit 27: forced on this small store; bit 2: no bitmap)
        for flags in (0, 2048, 32, 4, 16, 4096, 4096 | 256, 16384, 16384 | 256, 1 << 24, (1 << 24) | 256, 16384 | (1 << 25), 4 | (1 << 27)):
            eng.set_debug(flags)
            assert eng.ref_scan(case.hit_ratio, case.match_ratio, case.max_peak) == n_o     # clears the votes
            eng.vote()
            if flags & (1 << 27):
                assert eng.vote_info()["form"] == ("shared" if e <= 3 else "dense"), eng.vote_info()
            assert (eng.peaks_export(n_g)[1] == pf_o[:n_g]).all(), f"votes differ with debug flags {flags}"
            assert (eng.peaks_export(n_g)[0] == loci_o[:2 * n_o]).all() and (eng.peak_kmer_export() == pk_o).all(), flags
            form = eng.scan_info()["form"]
            fl = eng.flags_export(0, n_bases)
            assert (((fl ^ flags_g) & 0b1111100) == 0).all(), f"{form}: good / peak / inside / selected / new flags differ"
            exact = (fl & 0x80) != 0
            if flags & (1 << 24):
                assert form == "slot-first" and eng.slot_list()["entries"] > 0, (form, eng.slot_list())
            if form == "single-first":
                assert flags & 4096
                assert (((fl ^ flags_g) & 1) == 0).all(), "single-first: the single flag is exact everywhere"
                assert (((fl ^ flags_g) & 0b10)[exact] == 0).all(), "single-first: trio flag differs where it claims to be exact"
                assert (((fl & ~flags_g) & 0b10) == 0).all(), "single-first: trio flag is not a lower bound"
            elif form in ("trio-first", "slot-first"):
                if flags & 16384:
                    assert form == "trio-first", (form, flags)      # bit 14 alone: the trio-first KERNEL, list or no list
                assert (((fl ^ flags_g) & 0b10) == 0).all(), "trio-first: the trio flag is exact everywhere"
                assert (((fl ^ flags_g) & 1)[exact] == 0).all(), "trio-first: single flag differs where it claims to be exact"
                assert (((fl & ~flags_g) & 1) == 0).all(), "trio-first: single flag is not a lower bound"
                if flags & 16384:
                    assert eng.scan_info()["tiles_exact"] <= eng.scan_info()["tiles"]
            else:
                assert (((fl ^ flags_g) & 0b11) == 0).all()
        eng.set_debug(0)
        # D
        out = str(tmp_path / "interval.txt")
        eng.write_intervals(out)
        assert open(out).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()
        # the other resident forms of the reference: the bases as bit-planes with the hashes recomputed in the scan (packed), and
        # the hashes computed straight from the FASTA without the index file -- every form of B1, the same oracle answers
        index_bytes = eng.reference_info()["resident_bytes"]
        for packed in (True, False):
            eng.set_reference_form(packed)
            assert eng.reference_info()["resident_bytes"] == 0
            if packed:
                with pytest.raises(RuntimeError):
                    eng.index_load(index)                 # hashes cannot be turned back into bases
            assert eng.reference_load_fasta(fa2) == (n_contigs, n_bases)
            info = eng.reference_info()
            assert info["form"] == ("packed" if packed else "index")
            assert info["resident_bytes"] == (12 * ((n_bases + 31) // 32 + 2) if packed else index_bytes)
            assert eng.slot_list()["entries"] == 0, "a new reference drops the list of the one before"
            # bit 24: the forms that read the slot list -- trio-first's (slot-first: either resident form) and, with bit 12, single-first's
            # (slot-single: packed form only; the list is rebuilt under the positions' smallest hash)
            for flags in (8192, 0, 4096, 4096 | 256, 16384, 16384 | 256, 1 << 24, (1 << 24) | 256, 0, 4096 | (1 << 24), 4096 | (1 << 24) | 256, 1 << 24):
                eng.set_debug(flags)
                assert eng.ref_scan(case.hit_ratio, case.match_ratio, case.max_peak) == n_o
                if flags & (1 << 24):
                    want_form = "slot-first" if not flags & 4096 else "slot-single" if packed else "single-first"
                    assert eng.scan_info()["form"] == want_form and (eng.slot_list()["entries"] > 0 or want_form == "single-first"), (flags, eng.scan_info())
                eng.vote()
                loci_p, pf_p = eng.peaks_export(n_g)
                assert (loci_p == loci_o[:2 * n_o]).all() and (pf_p == pf_o[:n_g]).all() and (eng.peak_kmer_export() == pk_o).all(), (packed, flags)
                fl = eng.flags_export(0, n_bases)
                assert (((fl ^ flags_g) & (0b1111111 if flags == 8192 else 0b1111100)) == 0).all(), (packed, flags)
                if eng.scan_info()["form"] == "slot-single":
                    assert (((fl ^ flags_g) & 1) == 0).all(), "slot-single: the single flag is exact everywhere"
                    assert (((fl ^ flags_g) & 0b10)[(fl & 0x80) != 0] == 0).all() and (((fl & ~flags_g) & 0b10) == 0).all(), "slot-single: trio flag exact where claimed, a lower bound elsewhere"
            eng.set_debug(0)
            eng.write_intervals(out)
            assert open(out).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()


# ------------------------------------------------------------------ the 12-argument contract against the reference goldens
@pytest.mark.parametrize("ref_form", ["index", "packed"])
@pytest.mark.parametrize("name", list(cases.CASES))
def test_extract_ref_matches_reference_golden(case_inputs, name, ref_form, tmp_path):
    """ref_form "packed": the bases resident instead of the index file's hashes -- the same files out, and the 12-bytes-per-base
    index is neither read (beyond its coder header, when the golden was made with the index in place) nor written"""
    from localhgt_amd import extract_ref, get_bed_file
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    interval = str(tmp_path / "interval.txt")
    argv = cases.extract_ref_argv(case, f1, f2, fa2, interval)
    index = f"{fa2}.k{case.k}.h{case.e}.index.dat"
    forms = (["index"] if case.preexisting_index else []) + [ref_form]      # a golden made with the index already in place (quirk Q3)
    for form in forms:
        rep = extract_ref.run(extract_ref.parse_argv(argv), log=lambda *a: None, emulate_threads=case.threads > 1, ref_form=form)
    gold = os.path.join(cases.GOLDEN_DIR, name)
    assert rep["n_peaks"] == meta["raw_peaks"] and rep["ref_form"] == ref_form
    assert open(interval).read() == open(os.path.join(gold, "interval.txt")).read()
    assert open(fa2 + ".genome.len.txt").read() == open(os.path.join(gold, "genome.len.txt")).read()
    if ref_form == "packed" and not case.preexisting_index:
        assert not os.path.exists(index)
        assert rep["ref_resident_bytes"] < 0.2 * rep["n_bases"] * 4 * case.e + 64
    else:
        assert cases.sha256_file(index) == meta["sha256"]["index.dat"]
    if case.bed_defined:
        n = get_bed_file.write_bed(fa2, interval)
        assert open(interval + ".bed").read() == open(os.path.join(gold, "interval.txt.bed")).read()
        assert f"extracted ref length is: {n}\n" == meta["bed_stdout"]
    else:
        with pytest.raises(get_bed_file.InconsistentReferenceIds):
            get_bed_file.write_bed(fa2, interval)


def test_shared_vote_with_no_room_for_its_events(case_inputs, tmp_path, monkeypatch):
    """the shared-line-fill vote (k_vote_shared.hip) keeps the offsets with a hit in an arena; with an arena of four blocks nearly
    every pair overflows and is voted by the generic kernel behind it: the reference's files all the same"""
    from localhgt_amd import extract_ref
    monkeypatch.setenv("LHGT_SHARED_ARENA", "0.0001")
    monkeypatch.setenv("LHGT_DEBUG", str(4 | (1 << 27)))
    for name in ("k24_base", "k24_t4", "k32_base"):
        case = cases.CASES[name]
        fa, f1, f2, meta = case_inputs(name)
        d = tmp_path / name
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        interval = str(d / "interval.txt")
        rep = extract_ref.run(extract_ref.parse_argv(cases.extract_ref_argv(case, f1, f2, fa2, interval)), log=lambda *a: None)
        assert rep["n_peaks"] == meta["raw_peaks"]
        assert open(interval).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read(), name


# ------------------------------------------------------------------ edge cases
def _pairs(reads1, reads2):
    s1 = np.frombuffer(b"".join(reads1), dtype=np.uint8)
    s2 = np.frombuffer(b"".join(reads2), dtype=np.uint8)
    o1 = np.cumsum([0] + [len(r) for r in reads1]).astype(np.uint64)
    o2 = np.cumsum([0] + [len(r) for r in reads2]).astype(np.uint64)
    return s1, o1, s2, o2


def _write_fq(path, reads, suffix):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@e%d/%s\n" % (i, suffix) + r + b"\n+\n" + b"I" * len(r) + b"\n")


def test_ragged_and_degenerate_reads(Engine, oracle, tmp_path):
    """reads shorter than k, exactly k, all-N, maximum length 500, empty; mates of unequal length"""
    k, e = 24, 3
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n):
        return acgt[rng.integers(0, 4, n)].tobytes()
    base = rnd(600)
    reads1 = [b"", rnd(5), rnd(23), base[:24], b"N" * 100, base[:500], rnd(150), base[10:160], b"A" * 150, rnd(33)]
    reads2 = [rnd(150), b"", rnd(24), base[100:124], rnd(100), base[100:600], b"n" * 30, base[10:160], b"T" * 150, rnd(32)]
    f1, f2 = str(tmp_path / "e.1.fq"), str(tmp_path / "e.2.fq")
    _write_fq(f1, reads1, b"1")
    _write_fq(f2, reads2, b"2")
    with Engine(k, e) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        cc = eng.coder_get()
        eng.sampling_init(100.0)
        eng.pairs_append(*_pairs(reads1, reads2))
        eng.count_kmers()
        got = eng.counts_export()
        table = np.zeros(1 << k, dtype=np.uint8)
        big = 1 << 40
        oracle.count(f1, big, k, e, cc, 100.0, None, table)
        oracle.count(f2, big, k, e, cc, 100.0, None, table)
        assert (got == table).all()
        assert got.max() == 3     # poly-A / its reverse complement poly-T saturate one slot per hash
        # same through the FASTQ parser
        eng.counts_clear()
        eng.pairs_clear()
        assert eng.pairs_load_fastq(f1, f2, 100.0) == (10, 10)
        eng.count_kmers()
        assert (eng.counts_export() == table).all()


def test_empty_inputs(Engine, tmp_path):
    k, e = 24, 3
    f1, f2 = str(tmp_path / "z.1.fq"), str(tmp_path / "z.2.fq")
    open(f1, "w").close()
    open(f2, "w").close()
    fa = str(tmp_path / "ref.fa")
    rng = np.random.default_rng(3)
    with open(fa, "w") as f:
        f.write(">c1 desc\n" + "".join("ACGT"[i] for i in rng.integers(0, 4, 3000)) + "\n>tiny\nACGT\n")
    from localhgt_amd import extract_ref
    interval = str(tmp_path / "i.txt")
    rep = extract_ref.run(extract_ref.parse_argv([f1, f2, fa, interval, "0.1", "0.08", "1", str(k), "1000", str(e), "1", "1"]),
                          log=lambda *a: None)
    assert rep["pairs_kept"] == 0 and rep["n_peaks"] == 0
    assert open(interval).read() == "1\t1\t1\n"          # the sentinel line alone (E:522-524, 542)
    assert open(fa + ".genome.len.txt").read() == "c1\t1\t3000\t3000\n"


def test_counts_merge_is_saturating_add(Engine):
    k, e = 16, 3
    rng = np.random.default_rng(9)
    with Engine(k, e) as a, Engine(k, e) as b:
        for eng, seed in ((a, 1), (b, 2)):
            eng.rng_seed(7)
            eng.coder_generate()
            r = np.random.default_rng(seed)
            reads = [np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, 150)].tobytes() for _ in range(3000)]
            reads += [b"A" * 150] * 3
            eng.pairs_append(*_pairs(reads, reads))
            eng.count_kmers()
        ta, tb = a.counts_export(), b.counts_export()
        pb, nb = b.counts_buffer()
        half = nb // 2
        a.counts_merge(pb + half, half, nb - half)     # merge only the upper half, device pointer arithmetic
        got = a.counts_export()
        want = ta.copy()
        lo = (half * 4)                                # 4 slots per byte
        want[lo:] = np.minimum(3, ta[lo:].astype(int) + tb[lo:].astype(int))
        assert (got == want).all()
        assert (ta[lo:] != want[lo:]).any()


def test_too_many_peaks_is_reported(Engine, case_inputs, tmp_path):
    from localhgt_amd import extract_ref, _lib
    case = cases.CASES["k24_base"]
    fa, f1, f2, _ = case_inputs("k24_base")
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    argv = cases.extract_ref_argv(case, f1, f2, fa2, str(tmp_path / "i.txt"))
    argv[8] = "10"   # max_peak
    with pytest.raises(_lib.LocalHGTError) as ei:
        extract_ref.run(extract_ref.parse_argv(argv), log=lambda *a: None)
    assert ei.value.code == 6 and "Too many peaks" in str(ei.value)


@pytest.mark.parametrize("k", [12, 18, 20, 25, 26, 32])
def test_partitioned_count_equals_direct_count(Engine, k):
    """the radix-partitioned phase A (default) and the direct CAS kernel fill the same table;
    k sweeps the partition geometry: no partition (k<=18), one level (<=25), two levels"""
    e = 3
    rng = np.random.default_rng(k)
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    reads1 = [acgt[rng.choice(5, size=int(rng.integers(0, 260)), p=[.248, .248, .248, .248, .008])].tobytes() for _ in range(6000)]
    reads2 = [acgt[rng.choice(5, size=int(rng.integers(0, 260)), p=[.248, .248, .248, .248, .008])].tobytes() for _ in range(6000)]
    reads1 += [b"A" * 200, b"ACAC" * 50] * 40          # hot slots
    reads2 += [b"T" * 200, b"GTGT" * 50] * 40
    c2 = (rng.random(len(reads1)) < 0.9).astype(np.uint8)
    tables = []
    for mode in (0, 1):
        with Engine(k, e) as eng:
            eng.rng_seed(11)
            eng.coder_generate()
            eng.set_count_mode(mode)
            eng.pairs_append(*_pairs(reads1, reads2), count_mate2=c2)
            eng.count_kmers()
            eng.count_kmers()          # twice: the second pass meets pre-filled slices
            tables.append(eng.counts_histogram() if k == 32 else eng.counts_export())
            if k == 32:
                tables[-1] = np.concatenate([tables[-1], eng.counts_export(0, 1 << 26).astype(np.uint64)])
    assert (tables[0] == tables[1]).all()


def test_direct_form_of_the_partition_equals_direct_count(Engine, monkeypatch):
    """round 4's scatters (k = 32, e = 3, reads of <= 159 bases: fixed-slot tiles, workgroup-private pieces, no histogram pass)
    against the compare-and-swap kernel and against round 3's sorted-tile scatters (LHGT_DEBUG bit 16): whole-table digest,
    histogram and the first 2^26 slots.  Hot k-mers (poly-A, poly-AC, one read repeated) overflow the 128 slots of a tile bucket,
    the 256 of the second level and the pieces themselves -- those keys go straight to the table; ragged lengths, N's, reads
    shorter than k, mates not counted (quirk Q4) and an odd number of reads exercise the tails.  The key scatter copies out whole
    128-byte lines (round 6: a bucket carries up to 63 keys from tile to tile); LHGT_PART_CG=8 is round 4's 16-byte copy-out."""
    k, e = 32, 3
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    def rd(n):
        return [acgt[rng.choice(5, size=int(rng.integers(0, 160)), p=[.248, .248, .248, .248, .008])].tobytes() for _ in range(n)]
    hot = acgt[rng.choice(4, size=150)].tobytes()
    reads1 = rd(20001) + [b"A" * 150, b"ACAC" * 37, hot] * 700 + rd(3000)
    reads2 = rd(20001) + [b"T" * 150, b"GTGT" * 37, hot] * 700 + rd(3000)
    c2 = (rng.random(len(reads1)) < 0.9).astype(np.uint8)
    got = []
    for mode, dbg, cg in ((0, 0, None), (1, 0, None), (1, 65536, None), (1, 1 << 21, None), (1, 0, "8")):      # bit 21: the Small geometry of the direct form
        if cg: monkeypatch.setenv("LHGT_PART_CG", cg)
        else: monkeypatch.delenv("LHGT_PART_CG", raising=False)
        with Engine(k, e) as eng:
            eng.rng_seed(11)
            eng.coder_generate()
            eng.set_count_mode(mode)
            eng.set_debug(dbg)
            eng.pairs_append(*_pairs(reads1, reads2), count_mate2=c2)
            eng.count_kmers()
            eng.count_kmers()          # twice: the second pass meets pre-filled slices
            got.append((eng.digest(eng.DIGEST_COUNTS), tuple(int(x) for x in eng.counts_histogram()), eng.counts_export(0, 1 << 26)))
    for other in got[1:]:
        assert other[0] == got[0][0] and other[1] == got[0][1] and (other[2] == got[0][2]).all()
    assert got[0][1][3] > 0 and sum(got[0][1]) == 1 << 32


def test_batch_of_long_reads_only_counts_like_the_direct_kernel(Engine):
    """a batch in which EVERY read has more than 128 k-mer offsets (250-base reads throughout): phase A's direct form skips its
    short-read launch and sends everything through the segment list (round 5) -- against the compare-and-swap kernel and the sorted
    tiles; ragged lengths 160..420, N's, a hot read, mates not counted"""
    k, e = 32, 3
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    def rd(n):
        return [acgt[rng.choice(5, size=int(rng.integers(160, 421)), p=[.248, .248, .248, .248, .008])].tobytes() for _ in range(n)]
    hot = acgt[rng.choice(4, size=250)].tobytes()
    reads1 = rd(9001) + [b"A" * 250, hot] * 300 + rd(1000)
    reads2 = rd(9001) + [b"T" * 250, hot] * 300 + rd(1000)
    c2 = (rng.random(len(reads1)) < 0.9).astype(np.uint8)
    got = []
    for mode, dbg in ((0, 0), (1, 0), (1, 65536), (1, 1 << 21)):
        with Engine(k, e) as eng:
            eng.rng_seed(11)
            eng.coder_generate()
            eng.set_count_mode(mode)
            eng.set_debug(dbg)
            eng.pairs_append(*_pairs(reads1, reads2), count_mate2=c2)
            eng.count_kmers()
            eng.count_kmers()
            got.append((eng.digest(eng.DIGEST_COUNTS), tuple(int(x) for x in eng.counts_histogram())))
    for other in got[1:]:
        assert other == got[0]
    assert got[0][1][3] > 0 and sum(got[0][1]) == 1 << 32


def test_more_voted_peaks_than_the_first_buffer_holds(Engine, tmp_path):
    """phase D compacts the voted peaks into a buffer that starts at 4096 records and grows once when more peaks were voted (round
    4: the grown buffer's 64-bit counter sat at an odd multiple of 12 bytes and the first sample with > 4096 voted peaks -- the
    SNP 1 % leg of bench.py -- faulted).  6000 short contigs covered four times over, whose ends are contrast peaks (E:931-932,
    644-671), and chimeric pairs that join the tail of one contig to the tail of the next vote for thousands of peaks: the
    interval file must hold exactly the merge of the peaks peaks_export reports as voted, on two runs in a row."""
    k, e = 24, 3
    rng = np.random.default_rng(3)
    nc, cl, L = 6000, 3000, 150
    ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=nc * cl)]
    offsets = (np.arange(nc + 1) * cl).astype(np.uint64)
    starts = np.arange(0, cl - L + 1, 37)
    contig0 = np.repeat(np.arange(nc), starts.size) * cl
    s1 = contig0 + np.tile(starts, nc)
    s2 = contig0 + np.tile(starts[::-1], nc)                  # the mate: another stretch of the same contig
    tail = np.arange(nc - 1) * cl + (cl - L)                  # chimeric: the last 150 bases of contig i with those of contig i + 1
    s1 = np.concatenate([s1, tail, tail]); s2 = np.concatenate([s2, tail + cl, tail + cl])
    idx = np.arange(L)
    m1, m2 = ref[s1[:, None] + idx].reshape(-1), ref[s2[:, None] + idx].reshape(-1)
    ro = (np.arange(s1.size + 1) * L).astype(np.uint64)
    with Engine(k, e) as eng:
        eng.rng_seed(5)
        eng.coder_generate()
        eng.index_from_memory(ref, offsets)
        eng.pairs_append(m1, ro, m2, ro)
        eng.count_kmers()
        n = eng.ref_scan(0.1, 0.08, 10**7)
        eng.vote()
        for rep in range(2):
            nf = eng.write_intervals(str(tmp_path / f"iv{rep}.txt"))
            loci, filt = eng.peaks_export(n)
            voted = np.flatnonzero(filt >= 1)
            assert nf == voted.size and nf > 4096 + 1, (n, nf, voted.size)
            # count_filtered_peak's merge (E:515-548) over the voted peaks in id order
            lines, chr_, start, end = [], 1, 1, 1
            for i in voted:
                c, pos = int(loci[2 * i]), int(loci[2 * i + 1])
                if chr_ == c and pos - 500 - end < 500:
                    end = pos + 500
                else:
                    lines.append(f"{chr_}\t{start}\t{end}\n")
                    chr_, start, end = c, pos - 500, pos + 500
            lines.append(f"{chr_}\t{start}\t{end}\n")
            assert open(tmp_path / f"iv{rep}.txt").read() == "".join(lines)


def test_count_diff_kmer_tool(oracle, case_inputs):
    """C-tool row: the printed rates are #(T==0)/2^k and #(T!=3)/2^k of phase A's table (count_diff_kmer.cpp:26-50)"""
    import io
    from localhgt_amd import count_diff_kmer
    fa, f1, f2, _ = case_inputs("k24_seed7")
    k = 20
    buf = io.StringIO()
    hist = count_diff_kmer.run(f1, f2, k, 100, seed=3, out=buf)
    oracle.srand(3)
    cc = oracle.random_coder(k, 3)
    table = np.zeros(1 << k, dtype=np.uint8)
    big = 1 << 40
    oracle.count(f1, big, k, 3, cc, 100.0, None, table)
    oracle.count(f2, big, k, 3, cc, 100.0, None, table)
    want = np.bincount(table, minlength=4)
    assert (hist == want).all()
    lines = buf.getvalue().splitlines()
    assert lines[0] == f"###kmer_is {k} sample_ratio_is 100"
    size = 1 << k
    assert lines[2] == "####%d\t%g\t%g" % (size, np.float32((size - want[3]) / size), np.float32(want[0] / size))


def test_vote_prefilter_changes_nothing(Engine):
    """the L2-resident folded bitmap in front of peak_kmer only skips probes that would return 0"""
    k, e = 28, 3
    with Engine(k, e) as eng:
        eng.rng_seed(5)
        eng.coder_generate()
        eng.synth_reference(3, 16, 100_000)
        eng.synth_pairs(3, 4, 16, 100_000, 0, 60_000)
        eng.count_kmers()
        votes = []
        for flags in (0, 16, 32, 4, 128, 256, 16 | 2048, (1 << 20) | 16, (1 << 20) | 32, (1 << 20) | 16 | 2048):   # bit 20: the 3 MiB form of the bitmap (queued kernel, generic kernel, queued kernel's direct branch)  # fold kernel (128 KiB, deferred judge), queued kernel, generic kernel with bitmap, no prefilter, scan variants, queued kernel's direct branch
            eng.set_debug(flags)
            n = eng.ref_scan(0.1, 0.08, 10**7)
            eng.vote()
            loci, v = eng.peaks_export(n)
            votes.append((n, loci.copy(), v.copy()))
        eng.set_debug(0)
    assert all(v[0] == votes[0][0] for v in votes) and votes[0][0] > 50
    for other in votes[1:]:
        assert (votes[0][1] == other[1]).all() and (votes[0][2] == other[2]).all()
    assert votes[0][2].max() >= 1


def test_reference_without_indexable_contig(tmp_path, oracle):
    """every contig <= k: the index holds only its header and the interval file only the sentinel line (E:542)"""
    from localhgt_amd import extract_ref
    fa = str(tmp_path / "ref.fa")
    open(fa, "w").write(">a\nACGTACGTAC\n>b\nGGGTTTAAACCC\n")
    f1, f2 = str(tmp_path / "s.1.fq"), str(tmp_path / "s.2.fq")
    _write_fq(f1, [b"ACGT" * 30], b"1")
    _write_fq(f2, [b"TTGCA" * 24], b"2")
    interval = str(tmp_path / "i.txt")
    args = [f1, f2, fa, interval, "0.1", "0.08", "1", "24", "1000", "3", "1", "1"]
    rep = extract_ref.run(extract_ref.parse_argv(args), log=lambda *a: None)
    assert rep["n_contigs"] == 0 and rep["n_peaks"] == 0 and rep["pairs_kept"] == 1
    assert open(interval).read() == "1\t1\t1\n"
    assert os.path.getsize(fa + ".k24.h3.index.dat") == 1200 and open(fa + ".genome.len.txt").read() == ""
    d = tmp_path / "o"
    d.mkdir()
    fa2 = str(d / "ref.fa")
    shutil.copy(fa, fa2)
    rc, _ = oracle.run(f1, f2, fa2, str(d / "i.txt"), 0.1, 0.08, 1, 24, 1000, 3, 1, 1.0)
    assert rc == 0 and open(d / "i.txt").read() == "1\t1\t1\n"
    assert open(fa2 + ".k24.h3.index.dat", "rb").read() == open(fa + ".k24.h3.index.dat", "rb").read()


def test_executables_as_pipeline_sh_calls_them(case_inputs, tmp_path):
    """scripts/pipeline.sh:35-36 verbatim, with this repo's bin/ on PATH and NO environment variable of ours: extract_ref <12 args>;
    get_bed_file.py ref interval > log.  -t 10 is what `localhgt bkp` passes by default (scripts/localhgt.py:52), and under
    sampling the reference's -t 10 keeps other reads than its -t 1 (E:1037): the files must be those of the reference's -t 10 run"""
    import subprocess
    import sys
    name = "k24_t10_sample_bases"
    case = cases.CASES[name]
    assert case.threads == 10
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("LHGT_")}
    env["PATH"] = os.path.join(root, "bin") + os.pathsep + os.environ["PATH"]
    interval = str(tmp_path / "S.interval.txt")
    script = (f'extract_ref {f1} {f2} {fa2} {interval} {case.hit_ratio} {case.match_ratio} 10 {case.k} {case.max_peak} {case.e} {case.seed} {int(case.sample)}\n'
              f'python3 {root}/bin/get_bed_file.py {fa2} {interval} > {tmp_path}/S.log\n')
    res = subprocess.run(["bash", "-c", script], env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    gold = os.path.join(cases.GOLDEN_DIR, name)
    assert open(interval).read() == open(os.path.join(gold, "interval.txt")).read()
    assert open(interval + ".bed").read() == open(os.path.join(gold, "interval.txt.bed")).read()
    assert open(tmp_path / "S.log").read() == meta["bed_stdout"]
    assert cases.sha256_file(f"{fa2}.k{case.k}.h{case.e}.index.dat") == meta["sha256"]["index.dat"]


def test_loader_survives_bad_input(Engine, case_inputs, tmp_path):
    """a load that fails half-way (a read of 600 bases in the middle of the file) or at once (a second file of foreign reads) returns its
    error, leaves nothing resident, and the same engine then loads good files as if nothing had happened"""
    from localhgt_amd import _lib
    fa, f1, f2, _ = case_inputs("k24_base")
    raw1 = open(f1, "rb").read().split(b"\n")
    bad1 = str(tmp_path / "long.1.fq")
    mid = (len(raw1) // 8) * 4 + 1
    open(bad1, "wb").write(b"\n".join(raw1[:mid] + [b"A" * 600] + raw1[mid + 1:]))
    short2 = str(tmp_path / "foreign.2.fq")       # no line carries fq1's first read ID: the reference would spin through 10^9 failed reads
    open(short2, "wb").write(open(f2, "rb").read().replace(b"@r", b"@q"))
    with Engine(24, 3) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        want = None
        for attempt in range(2):
            for a, b in ((bad1, f2), (f1, short2)):
                with pytest.raises(_lib.LocalHGTError) as ei:
                    eng.pairs_load_fastq(a, b, 100.0)
                assert ei.value.code == 4
                eng.pairs_clear()
            eng.counts_clear()
            seen, kept = eng.pairs_load_fastq(f1, f2, 100.0)
            assert seen == kept == eng.pairs_count() > 1000
            eng.count_kmers()
            got = eng.counts_export().tobytes()
            want = want or got
            assert got == want
            eng.pairs_clear()


def test_loader_without_page_locked_memory(case_inputs, tmp_path, monkeypatch):
    """the FASTQ pipeline's fallback when the host refuses pinned memory (LHGT_NO_PINNED forces it): pageable buffers, same files"""
    from localhgt_amd import extract_ref
    monkeypatch.setenv("LHGT_NO_PINNED", "1")
    for name in ("k24_sample_half_fresh", "k24_t4"):
        case = cases.CASES[name]
        fa, f1, f2, meta = case_inputs(name)
        d = tmp_path / name
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        interval = str(d / "interval.txt")
        rep = extract_ref.run(extract_ref.parse_argv(cases.extract_ref_argv(case, f1, f2, fa2, interval)), log=lambda *a: None,
                              emulate_threads=case.threads > 1)
        assert rep["n_peaks"] == meta["raw_peaks"]
        assert open(interval).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()


def test_extract_ref_executable_honours_t_when_asked(case_inputs, tmp_path):
    """`localhgt bkp` passes -t 10 by default: bin/extract_ref gives the reference's -t 10 file (golden from the reference run with
    its threads in creation order); LHGT_EMULATE_THREADS=0 gives the -t 1 file whatever -t says"""
    import subprocess
    name = "k24_t10_sample_bases"
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, extra in (("emulated", {}), ("plain", {"LHGT_EMULATE_THREADS": "0"})):
        d = tmp_path / tag
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        interval = str(d / "S.interval.txt")
        env = {k: v for k, v in os.environ.items() if k != "LHGT_EMULATE_THREADS"}
        env.update(PATH=os.path.join(root, "bin") + os.pathsep + os.environ["PATH"], **extra)
        num = str(int(case.sample)) if float(case.sample) == int(case.sample) else repr(float(case.sample))
        res = subprocess.run(["extract_ref", f1, f2, fa2, interval, repr(case.hit_ratio), repr(case.match_ratio), str(case.threads), str(case.k),
                              str(case.max_peak), str(case.e), str(case.seed), num], env=env, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        outs[tag] = open(interval).read()
    gold = open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()
    assert outs["emulated"] == gold
    assert outs["plain"] != gold and outs["plain"].count("1\t1\t1\n") == 1      # one thread range, one sentinel line
    assert gold.count("1\t1\t1\n") >= 8                                           # ten ranges, most of them empty


def test_thread_emulation_falls_back_to_t1_where_the_reference_run_is_undefined(oracle, case_inputs, tmp_path):
    """inputs on which the reference's -t N reads stale bytes (a thread entering a FASTQ within 1000 bytes of its end) or lets a
    thread's peaks run into the next thread's ids: one warning line, then the -t 1 result (golden / oracle)"""
    from localhgt_amd import extract_ref
    case = cases.CASES["k24_base"]
    fa, f1, f2, meta = case_inputs("k24_base")
    gold = open(os.path.join(cases.GOLDEN_DIR, "k24_base", "interval.txt")).read()
    # (1) a thread's id range overflows: 4 threads x 50 ids for 167 peaks in two contig groups
    d = tmp_path / "overflow"
    d.mkdir()
    fa2 = str(d / "ref.fa")
    shutil.copy(fa, fa2)
    interval = str(d / "i.txt")
    argv = cases.extract_ref_argv(case, f1, f2, fa2, interval)
    argv[6], argv[8] = "4", "200"
    log = []
    rep = extract_ref.run(extract_ref.parse_argv(argv), log=lambda *a: log.append(" ".join(str(x) for x in a)))
    assert rep["emulated_threads"] == 1 and rep["n_peaks"] == meta["raw_peaks"] and open(interval).read() == gold
    warn = [ln for ln in log if ln.startswith("warning")]
    assert len(warn) == 1 and "Too many peaks! thread" in warn[0] and "-t 1 result" in warn[0]
    # (2) 40 threads on six records: chunks near the end of the file
    d = tmp_path / "tiny"
    d.mkdir()
    t1, t2 = str(d / "t.1.fq"), str(d / "t.2.fq")
    open(t1, "wb").write(b"\n".join(open(f1, "rb").read().split(b"\n")[:24]) + b"\n")
    open(t2, "wb").write(b"\n".join(open(f2, "rb").read().split(b"\n")[:24]) + b"\n")
    outs = {}
    for who in ("gpu", "cpu"):
        w = d / who
        w.mkdir()
        fa3 = str(w / "ref.fa")
        shutil.copy(fa, fa3)
        iv = str(w / "i.txt")
        if who == "gpu":
            log = []
            rep = extract_ref.run(extract_ref.parse_argv([t1, t2, fa3, iv, "0.1", "0.08", "40", "24", "100000", "3", "1", "1"]),
                                  log=lambda *a: log.append(" ".join(str(x) for x in a)))
            assert rep["emulated_threads"] == 1 and sum(ln.startswith("warning") for ln in log) == 1
        else:
            rc, _ = oracle.run(t1, t2, fa3, iv, 0.1, 0.08, 1, 24, 100000, 3, 1, 1.0)
            assert rc == 0
        outs[who] = open(iv).read()
    assert outs["gpu"] == outs["cpu"]


@pytest.mark.parametrize("k,e", [(20, 9), (33 - 1, 1), (16, 4)])
def test_whole_run_matches_oracle_for_unusual_e(oracle, case_inputs, tmp_path, k, e):
    """e = 9 (four rand() rows per position, 9th hash outside the 8-bit nzmask), e = 1 and e = 4: GPU run vs oracle run"""
    from localhgt_amd import extract_ref
    fa, f1, f2, _ = case_inputs("k24_seed7")
    outs = {}
    for who in ("gpu", "cpu"):
        d = tmp_path / who
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        interval = str(d / "interval.txt")
        if who == "gpu":
            rep = extract_ref.run(extract_ref.parse_argv([f1, f2, fa2, interval, "0.1", "0.02", "1", str(k), "100000", str(e), "5", "0.7"]),
                                  log=lambda *a: None)
        else:
            rc, orep = oracle.run(f1, f2, fa2, interval, 0.1, 0.02, 1, k, 100000, e, 5, 0.7)
            assert rc == 0
        outs[who] = (open(interval).read(), open(fa2 + ".genome.len.txt").read(), cases.sha256_file(f"{fa2}.k{k}.h{e}.index.dat"))
    assert outs["gpu"] == outs["cpu"]
    assert rep["n_peaks"] == orep.n_peaks and rep["pairs_kept"] == orep.pairs_voted


def test_saturated_table_line_summary(Engine, oracle, case_inputs, tmp_path):
    """a table that the reads saturate (k = 16): ref_flags answers from the saturated-line bitmap; same flags, peaks and
    interval file as without it and as the oracle"""
    from localhgt_amd import extract_ref
    fa, f1, f2, _ = case_inputs("k24_base")
    k, e = 16, 3
    outs = {}
    for who in ("gpu", "cpu"):
        d = tmp_path / who
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        interval = str(d / "i.txt")
        if who == "gpu":
            extract_ref.run(extract_ref.parse_argv([f1, f2, fa2, interval, "0.1", "0.08", "1", str(k), "100000", str(e), "1", "1"]), log=lambda *a: None)
        else:
            assert oracle.run(f1, f2, fa2, interval, 0.1, 0.08, 1, k, 100000, e, 1, 1.0)[0] == 0
        outs[who] = open(interval).read()
    assert outs["gpu"] == outs["cpu"]
    index = str(tmp_path / "gpu" / f"ref.fa.k{k}.h{e}.index.dat")
    with Engine(k, e) as eng:
        _, n_bases = eng.index_load(index)
        eng.sampling_init(100.0)
        eng.pairs_load_fastq(f1, f2, 100.0)
        eng.count_kmers()
        hist = eng.counts_histogram()
        assert hist[3] > 0.9 * (1 << k), "the case is meant to saturate the table"
        res = []
        for flags in (0, 64, 128, 256, 384, 4096, 8192):   # + chunked tile scan, + no tile settled by window_good alone, lite / exact scan forced
            eng.set_debug(flags)
            n = eng.ref_scan(0.1, 0.08, 100000)
            res.append((n, eng.flags_export(0, n_bases) & 0b1111101, eng.peaks_export(n)[0].copy()))   # the trio bit is a lower bound after the lite form
        for other in res[1:]:
            assert res[0][0] == other[0] and (res[0][1] == other[1]).all() and (res[0][2] == other[2]).all()


# ------------------------------------------------------------------ count_diff_kmer --compat against the reference tool
def test_count_diff_kmer_compat_prints_the_reference_lines(oracle, case_inputs):
    """bin/count_diff_kmer --compat: the three result lines of the reference binary (time() fixed, threads in creation order;
    tests/golden/count_diff_kmer.json) and the whole table histogram of the oracle's restatement"""
    import io
    import json
    from localhgt_amd import count_diff_kmer
    fa, f1, f2, _ = case_inputs("k24_seed7")
    for g in json.load(open(os.path.join(cases.GOLDEN_DIR, "count_diff_kmer.json"))):
        buf = io.StringIO()
        hist = count_diff_kmer.run(f1, f2, g["k"], g["ratio"], out=buf, compat=True, compat_time=g["time"])
        assert buf.getvalue().splitlines() == g["lines"], g
        rc, want = oracle.count_diff_kmer(f1, f2, g["k"], g["ratio"], g["time"])
        assert rc == 0 and [int(x) for x in hist] == [int(x) for x in want]


def test_count_diff_kmer_compat_n_bases_and_overlap(oracle, tmp_path):
    """reads with N and other letters (never rejected: they code 1 on both strands, C:155-160), lower case, and a file small
    enough that every chunk overruns deep into the next one"""
    from localhgt_amd import count_diff_kmer
    rng = np.random.default_rng(3)
    f1, f2 = str(tmp_path / "n.1.fq"), str(tmp_path / "n.2.fq")
    for path, suf in ((f1, "1"), (f2, "2")):
        with open(path, "w") as f:
            for i in range(600):
                s = "".join(rng.choice(list("ACGTacgtNNRn"), size=90))
                f.write(f"@q{i}/{suf}\n{s}\n+\n{'I' * 90}\n")
    for k, ratio, t in ((14, 100, 1), (20, 55, 42)):
        hist = count_diff_kmer.run(f1, f2, k, ratio, out=open(os.devnull, "w"), compat=True, compat_time=t)
        rc, want = oracle.count_diff_kmer(f1, f2, k, ratio, t)
        assert rc == 0 and [int(x) for x in hist] == [int(x) for x in want]


@pytest.mark.parametrize("name,text", cases.odd_fastas(), ids=[n for n, _ in cases.odd_fastas()])
def test_odd_fasta_line_structure(Engine, oracle, tmp_path, name, text):
    """The FASTA never passes through a host parser: '>' lines and newline counts are found on the host, the text itself goes to
    the GPU and is stripped there (strip_fasta_block).  Index bytes and genome.len.txt against the restatement (itself pinned on
    these files against the reference binary, tests/test_oracle_vs_ref_fuzz.py), and the packed form loaded from the same text
    against the index form: same peaks, loci and peak_kmer."""
    k, e = 16, 3
    fa = str(tmp_path / "ref.fa")
    open(fa, "wb").write(text)
    rng = np.random.default_rng(len(text))
    body = bytes(c for c in text.upper() if c in b"ACGT")
    reads = [body[o:o + 120] for o in rng.integers(0, max(1, len(body) - 120), size=64)]
    f1, f2 = str(tmp_path / "s.1.fq"), str(tmp_path / "s.2.fq")
    _write_fq(f1, reads, b"1")
    _write_fq(f2, reads[::-1], b"2")
    with Engine(k, e) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        cc = eng.coder_get()
        n_contigs, n_bases = eng.index_build(fa, fa + ".idx", fa + ".len")
        assert oracle.index_build(fa, fa + ".oidx", fa + ".olen", k, e, cc) == n_contigs
        got, want = open(fa + ".idx", "rb").read(), open(fa + ".oidx", "rb").read()
        assert len(got) == len(want) and got[:1198] == want[:1198] and got[1200:] == want[1200:]
        assert open(fa + ".len").read() == open(fa + ".olen").read()
        res = {}
        for packed in (False, True):
            eng.set_reference_form(packed)
            if packed:
                assert eng.reference_load_fasta(fa, fa + ".plen") == (n_contigs, n_bases)
                assert open(fa + ".plen").read() == open(fa + ".olen").read()
            else:
                assert eng.index_load(fa + ".idx") == (n_contigs, n_bases)
            eng.pairs_clear()
            eng.sampling_init(100.0)
            eng.pairs_load_fastq(f1, f2, 100.0)
            eng.counts_clear()
            eng.count_kmers()
            eng.set_debug(8192)
            n = eng.ref_scan(0.1, 0.08, 100000)
            eng.set_debug(0)
            res[packed] = (n, eng.peaks_export(n)[0].tolist() if n > 0 else [], eng.peak_kmer_export().tobytes())
        assert res[True] == res[False]


def test_fasta_longer_than_one_upload_span(Engine, tmp_path):
    """A FASTA is uploaded in spans (64 Mi bases for the index build, 256 Mi for the packed form) whose page-locking runs one span
    ahead on a helper thread; a span's text starts inside the last locked page of its predecessor.  An index built over three
    spans and loaded again, and the packed form loaded over two spans, against the same bases generated on the device (index
    bytes against the restatement: test_odd_fasta_line_structure and the golden cases): same peaks, loci, registry and votes."""
    import bench
    k, e, cl = 32, 3, 1_000_000
    fa = str(tmp_path / "ref.fa")
    for packed, nc in ((False, 150), (True, 300)):
        res = {}
        for from_file in (False, True):
            with Engine(k, e) as eng:
                eng.rng_seed(1)
                eng.coder_generate()
                eng.set_reference_form(packed)
                ref = eng.synth_reference(1, nc, cl, want_host=from_file)
                if from_file:
                    bench.write_fasta(fa, ref, nc, cl)
                    del ref
                    if packed:
                        assert eng.reference_load_fasta(fa, None) == (nc, nc * cl)
                    else:
                        assert eng.index_build(fa, fa + ".idx", fa + ".len") == (nc, nc * cl)
                        assert eng.index_load(fa + ".idx") == (nc, nc * cl)
                        os.remove(fa + ".idx")
                eng.synth_options(0, 20, 30)
                eng.synth_pairs(1, 2, nc, cl, 0, 1_000_000, 150)
                eng.counts_clear()
                eng.count_kmers()
                n = eng.ref_scan(0.1, 0.08, 3_000_000)
                eng.vote()
                res[from_file] = (n, eng.digest(eng.DIGEST_LOCI), eng.digest(eng.DIGEST_PEAK_KMER), eng.digest(eng.DIGEST_VOTES))
        assert res[True] == res[False] and res[True][0] > 0, (packed, res)


def test_a_few_long_reads_do_not_demote_their_batch(Engine, oracle, tmp_path):
    """round 5: a batch of 150-base reads with a few longer ones (at most an eighth of its reads with more than 128 k-mer offsets)
    keeps the fast forms -- phase A's direct scatters take the long reads cut into segments of 128 k-mer offsets (also when every
    read is long), the queued / fold votes list the pairs with a long read for the generic kernel.  Same count table as the compare-and-swap kernel
    alone and as round 3's generic scatters; same votes as the generic vote without any filter; from files against the oracle."""
    k, e = 32, 3
    with Engine(k, e) as eng:
        eng.rng_seed(5)
        eng.coder_generate()
        eng.synth_reference(3, 40, 200_000)
        eng.synth_read_mix(20, 250)                         # 2 % of the pairs: 250-base reads
        eng.synth_options(0, 20, 8)
        eng.synth_pairs(3, 4, 40, 200_000, 0, 400_000)
        eng.synth_read_mix(0, 0)
        tables = []
        for mode, dbg in ((1, 0), (0, 0), (1, 65536)):      # partition (mixed form), compare-and-swap alone, round 3's generic scatters
            eng.set_count_mode(mode)
            eng.set_debug(dbg)
            eng.counts_clear()
            eng.count_kmers()
            tables.append((eng.digest(eng.DIGEST_COUNTS), tuple(int(x) for x in eng.counts_histogram())))
        eng.set_debug(0)
        eng.set_count_mode(-1)
        assert tables[0] == tables[1] == tables[2] and tables[0][1][3] > 1000
        # ... and batches of long reads only, of every length class up to the reference's 500 (ragged: segments of 1 .. 128 offsets)
        rng = np.random.default_rng(12)
        acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
        reads1 = [acgt[rng.choice(5, size=int(rng.integers(150, 501)), p=[.248, .248, .248, .248, .008])].tobytes() for _ in range(30_000)]
        reads2 = [acgt[rng.choice(5, size=int(rng.integers(0, 501)), p=[.248, .248, .248, .248, .008])].tobytes() for _ in range(30_000)]
        eng.pairs_clear()
        eng.pairs_append(*_pairs(reads1, reads2), count_mate2=(rng.random(len(reads1)) < 0.9).astype(np.uint8))
        ragged = []
        for mode, dbg in ((1, 0), (0, 0), (1, 65536)):
            eng.set_count_mode(mode)
            eng.set_debug(dbg)
            eng.counts_clear()
            eng.count_kmers()
            ragged.append((eng.digest(eng.DIGEST_COUNTS), tuple(int(x) for x in eng.counts_histogram())))
        eng.set_debug(0)
        eng.set_count_mode(-1)
        assert ragged[0] == ragged[1] == ragged[2]
        eng.pairs_clear()
        eng.counts_clear()
        eng.synth_read_mix(20, 250)
        eng.synth_pairs(3, 4, 40, 200_000, 0, 400_000)
        eng.synth_read_mix(0, 0)
        eng.count_kmers()
        votes = []
        for flags in (0, 16, 32, 4):                         # the form the engine picks (fold), queued, generic behind the bitmap, no filter at all
            eng.set_debug(flags)
            n = eng.ref_scan(0.1, 0.08, 10**7)
            eng.vote()
            loci, v = eng.peaks_export(n)
            votes.append((n, v.copy(), eng.vote_info()["form"]))
        eng.set_debug(0)
        assert votes[0][0] > 20 and votes[0][1].max() >= 1
        for other in votes[1:]:
            assert other[0] == votes[0][0] and (other[1] == votes[0][1]).all()
        assert {votes[0][2], votes[1][2]} <= {"fold", "queued"} and votes[0][2] != votes[2][2], [v[2] for v in votes]
        # 250-base reads throughout: the fold kernel's wide instance (four slices of 64 offsets per mate) against the generic kernels
        eng.pairs_clear()
        eng.counts_clear()
        eng.synth_pairs(3, 4, 40, 200_000, 0, 300_000, 250)
        eng.count_kmers()
        wide = []
        for flags in (0, 32, 4):
            eng.set_debug(flags)
            n = eng.ref_scan(0.1, 0.08, 10**7)
            eng.vote()
            wide.append((n, eng.peaks_export(n)[1].copy(), eng.vote_info()["form"]))
        eng.set_debug(0)
        assert wide[0][2] == "fold" and wide[1][2] == "bitmap" and wide[0][0] > 20 and wide[0][1].max() >= 1, [w[2] for w in wide]
        for other in wide[1:]:
            assert other[0] == wide[0][0] and (other[1] == wide[0][1]).all()
    # from files, against the oracle: k = 24 (phase C's path; phase A's applies at k = 32 and is covered above)
    k = 24
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = [acgt[rng.integers(0, 4, 60_000)] for _ in range(6)]
    # the sample: contig 1 with 3 kb of contig 2 pasted in, contig 2 without them -- both junction kinds
    sample = [np.concatenate([contigs[0][:30_000], contigs[1][20_000:23_000], contigs[0][30_000:]]),
              np.concatenate([contigs[1][:20_000], contigs[1][23_000:]])]
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">g%d\n" % (i + 1) + c.tobytes() + b"\n")
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    r1, r2 = [], []
    for i in range(9000):
        g = sample[i % 2]
        L = 250 if i % 50 == 0 else 150
        flen = int(rng.integers(300, 500))
        st = int(rng.integers(0, len(g) - flen))
        r1.append(g[st:st + L].tobytes())
        r2.append(comp[g[st + flen - L:st + flen]][::-1].tobytes())
    f1, f2 = str(tmp_path / "m.1.fq"), str(tmp_path / "m.2.fq")
    _write_fq(f1, r1, b"1")
    _write_fq(f2, r2, b"2")
    from localhgt_amd import extract_ref
    a = extract_ref.Args(f1, f2, fa, str(tmp_path / "gpu.txt"), 0.1, 0.08, 1, k, 100_000, 3, 1, 1.0)
    rep = extract_ref.run(a, log=lambda *x: None)
    fa2 = str(tmp_path / "cpu.fa")
    import shutil
    shutil.copy(fa, fa2)
    rc, orep = oracle.run(f1, f2, fa2, str(tmp_path / "cpu.txt"), 0.1, 0.08, 1, k, 100_000, 3, 1, 1.0)
    assert rc == 0 and (rep["n_peaks"], rep["n_filtered"]) == (orep.n_peaks, orep.n_filtered) and orep.n_filtered >= 1
    assert open(str(tmp_path / "gpu.txt")).read() == open(str(tmp_path / "cpu.txt")).read()


def test_single_pass_gives_up_midway_and_the_planned_loader_takes_over(Engine, oracle, case_inputs, tmp_path, monkeypatch):
    """the single-pass loader meets a line longer than a chunk's margin two thirds into the files, after batches have been installed
    AND counted (count-on-load; LHGT_INGEST_BATCH_PAIRS makes a small file close batches): what it delivered is taken back -- batches
    dropped, the count table cleared -- and the planned loader's result stands: same interval file as with the single pass switched
    off, and as the oracle's"""
    from localhgt_amd import extract_ref, _lib
    import ctypes as C
    fa, f1, f2, meta = case_inputs("k24_seed7")
    l1, l2 = open(f1, "rb").read().split(b"\n"), open(f2, "rb").read().split(b"\n")
    cut = (len(l1) * 2 // 3) // 4 * 4
    g1, g2 = str(tmp_path / "g.1.fq"), str(tmp_path / "g.2.fq")
    open(g1, "wb").write(b"\n".join(l1[:cut] + [b"@long/1", b"ACGT" * 30, b"+", b"I" * 70000] + l1[cut:]))
    open(g2, "wb").write(b"\n".join(l2[:cut] + [b"@long/2", b"ACGT" * 30, b"+", b"I" * 70000] + l2[cut:]))
    case = cases.CASES["k24_seed7"]
    monkeypatch.setenv("LHGT_INGEST_CHUNK_BYTES", "30000")
    monkeypatch.setenv("LHGT_INGEST_BATCH_PAIRS", "512")
    outs = {}
    for tag, stream in (("single pass first", "1"), ("planned only", "0")):
        monkeypatch.setenv("LHGT_INGEST_STREAM", stream)
        d = tmp_path / tag.replace(" ", "_")
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        a = extract_ref.Args(g1, g2, fa2, str(d / "i.txt"), case.hit_ratio, case.match_ratio, 1, case.k, case.max_peak, case.e, case.seed, 1.0)
        rep = extract_ref.run(a, log=lambda *x: None)
        why = C.create_string_buffer(200)
        path = _lib.load().lhgt_ingest_last_path(why, 200)
        outs[tag] = (open(str(d / "i.txt")).read(), rep["pairs_kept"], rep["n_peaks"], rep["n_filtered"], path, why.value.decode())
    assert outs["single pass first"][:4] == outs["planned only"][:4]
    assert outs["single pass first"][4] == 0 and "longer than a chunk's margin" in outs["single pass first"][5], outs["single pass first"][4:]
    d = tmp_path / "cpu"
    d.mkdir()
    fa2 = str(d / "ref.fa")
    shutil.copy(fa, fa2)
    rc, orep = oracle.run(g1, g2, fa2, str(d / "i.txt"), case.hit_ratio, case.match_ratio, 1, case.k, case.max_peak, case.e, case.seed, 1.0)
    assert rc == 0 and open(str(d / "i.txt")).read() == outs["planned only"][0] and orep.n_peaks == outs["planned only"][2] > 10
