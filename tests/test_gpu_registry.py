"""The peaks' k-mers registered BY PARTITION (round 6, k_scan.hip: rg_emit / rg_split / rg_apply; add_peak, E:247-267): routed by slot
through two scatter passes and applied per slice of peak_kmer in LDS, instead of one atomicMax per (slot, id) from the walk over the
reference.  "The larger id wins a slot" does not depend on the order, so peak_kmer, the loci, the votes and the interval files are
those of the direct kernel -- in one chunk and in several, with regions so tight that most records find them full and go to the table
at once, with and without the vote's bitmap, under the -t N emulation's id ranges."""
import os
import shutil

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
FORCE, NEVER = 1 << 29, 1 << 30


@pytest.mark.parametrize("variant", ["one_chunk", "three_chunks", "tight", "no_bitmap"])
def test_goldens_with_the_registry_by_partition(case_inputs, tmp_path, monkeypatch, variant):
    from localhgt_amd import extract_ref
    monkeypatch.setenv("LHGT_DEBUG", str(FORCE | (4 if variant == "no_bitmap" else 0)))
    if variant == "three_chunks":
        monkeypatch.setenv("LHGT_REGISTER_CHUNKS", "3")
    if variant == "tight":
        monkeypatch.setenv("LHGT_REGISTER_TIGHT", "40")
    if variant == "no_bitmap":
        monkeypatch.setenv("LHGT_VOTE_GROUPS", "2")          # ... and the dense vote collects its events without their contigs (k_vote.hip), -t N id ranges included
    for name in ("k24_base", "k24_t4", "k32_base", "k21_e3", "k20_e2", "k24_t10_sample_bases", "k24_nrun_lower"):
        case = cases.CASES[name]
        fa, f1, f2, meta = case_inputs(name)
        d = tmp_path / name
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        interval = str(d / "interval.txt")
        for ref_form in (["index"] if case.preexisting_index else []) + ["packed" if variant == "three_chunks" else "index"]:
            rep = extract_ref.run(extract_ref.parse_argv(cases.extract_ref_argv(case, f1, f2, fa2, interval)), log=lambda *a: None,
                                  emulate_threads=case.threads > 1, ref_form=ref_form)
        assert rep["n_peaks"] == meta["raw_peaks"]
        assert rep["registry_chunks"] == (3 if variant == "three_chunks" else 1), (name, rep["registry_chunks"])
        assert open(interval).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read(), (name, variant)


@pytest.mark.parametrize("k,e,packed", [(26, 3, True), (28, 2, False), (22, 3, True), (32, 3, True)])
def test_registry_by_partition_equals_the_direct_kernel(monkeypatch, k, e, packed):
    """a dense peak set (a sample at 0.3 x: nearly every read opens and closes a covered stretch, the regime of the CLI's default --sample)
    on a synthetic reference of 60 contigs: peak_kmer, loci and votes by digest, direct kernel against partition in 1 / 2 / 5 chunks and
    with tight regions"""
    from localhgt_amd.engine import Engine
    NC, CL, NP = 60, 50_000, 3_000
    want = None
    for variant in ("direct", "one", "two", "five", "tight", "tight_three"):
        monkeypatch.delenv("LHGT_REGISTER_CHUNKS", raising=False)
        monkeypatch.delenv("LHGT_REGISTER_TIGHT", raising=False)
        if variant in ("two", "five", "tight_three"):
            monkeypatch.setenv("LHGT_REGISTER_CHUNKS", {"two": "2", "five": "5", "tight_three": "3"}[variant])
        if variant.startswith("tight"):
            monkeypatch.setenv("LHGT_REGISTER_TIGHT", "25")
        with Engine(k, e) as eng:
            eng.set_debug(4 | (NEVER if variant == "direct" else FORCE))
            eng.rng_seed(1)
            eng.coder_generate()
            eng.set_reference_form(packed)
            eng.synth_reference(1, NC, CL)
            eng.synth_pairs(1, 5, NC, CL, 0, NP)
            eng.count_kmers()
            n = eng.ref_scan(0.1, 0.08, 10_000_000)
            info = eng.registry_info()
            eng.vote()
            got = (n, eng.digest(eng.DIGEST_PEAK_KMER), eng.digest(eng.DIGEST_LOCI), eng.digest(eng.DIGEST_VOTES))
        if variant == "direct":
            assert info["chunks"] == 0 and n > 500
            want = got
            continue
        assert info["chunks"] == {"one": 1, "two": 2, "five": 5, "tight": 1, "tight_three": 3}[variant], info
        assert info["records_bound"] > 0
        if variant.startswith("tight"):
            assert info["records_direct"] > info["records_bound"] // 4, info       # most records found their region full
        else:
            assert info["records_direct"] == 0, info
        assert got == want, (variant, got, want)


def test_the_direct_kernel_stays_where_the_partition_does_not_pay():
    """no forcing: a small peak set keeps the direct kernel (and its bitmap)"""
    from localhgt_amd.engine import Engine
    with Engine(26, 3) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        eng.synth_reference(1, 20, 50_000)
        eng.synth_pairs(1, 5, 20, 50_000, 0, 50_000)
        eng.count_kmers()
        eng.ref_scan(0.1, 0.08, 10_000_000)
        assert eng.registry_info()["chunks"] == 0


@pytest.mark.parametrize("k,e,pairs", [(26, 3, 3_000), (22, 3, 20_000), (24, 2, 6_000), (26, 3, 60_000)])
def test_dense_vote_bounds(monkeypatch, k, e, pairs):
    """round 6 (k_vote.hip): the dense vote walks a pair's events only if no bound clears it.  First over groups of whole contigs -- a
    peak's group follows from its id, so no contig is fetched until one group reaches six, and then only that group's --, then over
    hashed contigs with the read's own contig counted exactly and taken out.  Same votes with the group bound off
    (LHGT_VOTE_GROUPS=0) and with every bound off (debug bit 19), on dense peak sets where every read has dozens of events"""
    from localhgt_amd.engine import Engine
    NC, CL = 60, 50_000
    got = {}
    for variant, dbg in (("groups", 4), ("hashed", 4), ("walk", 4 | (1 << 19))):
        monkeypatch.setenv("LHGT_VOTE_GROUPS", "0" if variant == "hashed" else "2")       # 2: whatever share of the slots is registered
        with Engine(k, e) as eng:
            eng.set_debug(dbg)
            eng.rng_seed(1)
            eng.coder_generate()
            eng.set_reference_form(True)
            eng.synth_reference(1, NC, CL)
            eng.synth_pairs(1, 5, NC, CL, 0, pairs)
            eng.count_kmers()
            n = eng.ref_scan(0.1, 0.08, 10_000_000)
            eng.vote()
            assert eng.vote_info()["form"] == "dense"
            got[variant] = (n, eng.digest(eng.DIGEST_VOTES))
    assert got["groups"] == got["walk"] and got["hashed"] == got["walk"], got
    assert got["walk"][0] > 100
    print(k, e, pairs, got["walk"])


@pytest.mark.parametrize("nc,cl,pairs,threads,ragged", [(60, 50_000, 3_000, 1, 0), (700, 16_384, 12_000, 1, 0), (60, 50_000, 3_000, 1, 2_000), (60, 50_000, 3_000, 4, 0)])
def test_vote_groups_are_runs_of_whole_contigs(nc, cl, pairs, threads, ragged):
    """the invariant the dense vote's first bound stands on (k_vote.hip, k_scan.hip: vote_group_bounds): every group bound is the first
    peak id of a contig -- no contig's peaks lie in two groups --, the bounds ascend, and the groups' shares of the peaks are about
    equal where contigs allow.  Against the loci of all peaks (contig of every id), with many small contigs, few large ones, and the -t N
    emulation's id ranges"""
    from localhgt_amd.engine import Engine
    with Engine(26, 3) as eng:
        eng.set_debug(4)
        eng.rng_seed(1)
        eng.coder_generate()
        eng.set_reference_form(True)
        if ragged:          # the same base stream cut into ~2000 contigs of 30 .. 5000 bases (some no longer than k: they have no tiles)
            rng = np.random.default_rng(9)
            cuts = np.unique(np.concatenate([[0, nc * cl], rng.integers(1, nc * cl, ragged)])).astype(np.uint64)
            eng.synth_reference_cuts(1, nc, cl, cuts)
        else:
            eng.synth_reference(1, nc, cl)
        if threads > 1:
            eng.set_thread_emulation(threads)
        eng.synth_pairs(1, 5, nc, cl, 0, pairs)
        eng.count_kmers()
        n = eng.ref_scan(0.1, 0.08, 10_000_000)
        assert n > 300
        b = eng.vote_groups_export()
        assert b is not None and b[0] == 0 and b[1024] == 0xffffffff
        loci, _ = eng.peaks_export(n)
        chrs = loci[0::2].astype(np.int64)
        inner = b[1:1024].astype(np.int64)
        assert (np.diff(inner) >= 0).all()
        real = inner[inner < len(chrs)]
        assert len(real) > 0
        # a bound is where the contig changes (ids below the first real peak belong to nobody: contig 0)
        assert ((real == 0) | (chrs[real] != chrs[np.maximum(real - 1, 0)])).all()
        # ... and every contig's ids are one run, so "changes" means "a contig starts"
        starts = np.flatnonzero(np.diff(chrs) != 0) + 1
        seen = chrs[np.concatenate([[0], starts])]
        seen = seen[seen != 0]
        assert len(seen) == len(set(seen.tolist())), "a contig's peak ids are not one run"
