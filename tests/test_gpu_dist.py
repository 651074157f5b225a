"""GPU side of the multi-GPU path on ONE GPU: read sharding by blocks of the global pair ordinal,
the device merge kernel and the zero-copy torch views used for RCCL (world_size 1 over nccl)."""
import os
import shutil

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def test_sharded_engines_merge_to_the_unsharded_result(case_inputs, tmp_path):
    """two 'ranks' (run one after the other) each count their read shard; merging the tables and summing the
    votes gives the single-GPU result, and the interval file equals the reference golden"""
    from localhgt_amd.engine import Engine
    name = "k24_sample_half_cached"      # sampling active: decisions must not depend on the shard
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    k, e = case.k, case.e
    index = f"{fa2}.k{k}.h{e}.index.dat"
    with Engine(k, e) as b:      # build the index first so every engine sees a cached one (as in the golden run)
        b.rng_seed(case.seed)
        b.coder_generate()
        b.index_build(fa2, index, fa2 + ".genome.len.txt")
    engs = []
    world = 2
    for rank in range(world):
        eng = Engine(k, e)
        eng.rng_seed(case.seed)
        ratio = eng.sam_ratio(f1, case.sample)
        eng.index_load(index)
        eng.sampling_init(ratio)
        seen, kept = eng.pairs_load_fastq(f1, f2, ratio, rank, world, block=64)
        eng.count_kmers()
        engs.append((eng, kept))
    (e0, k0), (e1, k1) = engs
    with Engine(k, e) as whole:
        whole.rng_seed(case.seed)
        whole.index_load(index)
        whole.sampling_init(ratio)
        _, kept_all = whole.pairs_load_fastq(f1, f2, ratio)
        whole.count_kmers()
        assert k0 + k1 == kept_all and k0 > 0 and k1 > 0
        p1, n1 = e1.counts_buffer()
        p0, n0 = e0.counts_buffer()
        t0, t1 = e0.counts_export(), e1.counts_export()
        import torch
        from localhgt_amd.dist import device_tensor
        t0_dev = device_tensor(p0, n0, 0).clone()          # rank 0's table before it is merged into
        e0.counts_merge(p1, 0, n1)
        e1.counts_merge(t0_dev.data_ptr(), 0, n0)
        torch.cuda.synchronize()
        assert (e0.counts_export() == whole.counts_export()).all()
        assert (e0.counts_export() == np.minimum(3, t0.astype(int) + t1.astype(int))).all()
        assert (e1.counts_export() == e0.counts_export()).all()
        n = [x.ref_scan(case.hit_ratio, case.match_ratio, case.max_peak) for x in (e0, e1, whole)]
        assert n[0] == n[1] == n[2] == meta["raw_peaks"]
        for x in (e0, e1, whole):
            x.vote()
        f0, f1v, fw = (x.peaks_export(n[0])[1].astype(int) for x in (e0, e1, whole))
        assert (np.minimum(254, f0 + f1v) == fw).all()
    for eng, _ in engs:
        eng.close()


def test_exchange_world1_nccl_on_device_buffers(case_inputs, tmp_path):
    """the torch views of the raw device buffers are zero-copy and the collectives run on them"""
    import torch
    from localhgt_amd.dist import Exchange, device_tensor
    from localhgt_amd.engine import Engine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    ex = Exchange.from_env(backend="nccl")
    name = "k24_seed7"
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    try:
        with Engine(case.k, case.e) as eng:
            eng.rng_seed(case.seed)
            eng.coder_generate()
            index = f"{fa2}.k{case.k}.h{case.e}.index.dat"
            eng.index_build(fa2, index, fa2 + ".genome.len.txt")
            eng.index_load(index)
            eng.sampling_init(100.0)
            eng.pairs_load_fastq(f1, f2, 100.0)
            eng.count_kmers()
            before = eng.counts_export()
            p, n = eng.counts_buffer()
            view = device_tensor(p, n, 0)
            assert view.data_ptr() == p and view.numel() == n and view.dtype == torch.uint8
            ex.merge_counts(eng)
            assert (eng.counts_export() == before).all()
            n_peaks = eng.ref_scan(case.hit_ratio, case.match_ratio, case.max_peak)
            eng.vote()
            votes = eng.peaks_export(n_peaks)[1].copy()
            ex.sum_votes(eng)
            assert (eng.peaks_export(n_peaks)[1] == votes).all()
            out = str(tmp_path / "interval.txt")
            eng.write_intervals(out)
            assert open(out).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()
    finally:
        ex.close()


def _sharded_scan_sequential(engs, hit, match, max_peak):
    """the exchange of Exchange.sharded_scan done by hand for 'ranks' that live one after the other on one GPU"""
    import torch
    from localhgt_amd.dist import device_tensor
    counts = [e.ref_scan_local(hit, match) for e in engs]
    news = [c[0] for c in counts]
    n_total, n_sel = sum(news), sum(c[1] for c in counts)
    loci_parts, regs_parts = [], []
    for r, e in enumerate(engs):
        pl, pr, n_regs = e.ref_scan_emit(sum(news[:r]))
        loci_parts.append(device_tensor(pl, 8 * news[r], 0).view(torch.int32).clone() if news[r] else torch.empty(0, dtype=torch.int32, device="cuda:0"))
        regs_parts.append(device_tensor(pr, 8 * n_regs, 0).view(torch.int32).clone() if n_regs else torch.empty(0, dtype=torch.int32, device="cuda:0"))
    loci_all, regs_all = torch.cat(loci_parts), torch.cat(regs_parts)
    torch.cuda.synchronize()
    for e in engs:
        e.peaks_install(n_total, n_sel, max_peak, loci_all.data_ptr(), regs_all.data_ptr(), regs_all.numel() // 2)
    return n_total, news


@pytest.mark.parametrize("name,world", [("k24_base", 3), ("k24_nrun_lower", 2), ("k32_base", 4), ("k24_seed7", 8)])
def test_reference_sharded_scan_equals_whole_scan(case_inputs, tmp_path, name, world):
    """each 'rank' holds a contig shard of the index and the full count table; after the record exchange every rank's
    peak tables equal those of an unsharded scan, and the interval file equals the reference golden"""
    from localhgt_amd.engine import Engine
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    k, e = case.k, case.e
    index = f"{fa2}.k{k}.h{e}.index.dat"
    with Engine(k, e) as whole:
        whole.rng_seed(case.seed)
        whole.coder_generate()
        whole.index_build(fa2, index, fa2 + ".genome.len.txt")
        nc_all, nb_all = whole.index_load(index)
        whole.sampling_init(100.0)
        whole.pairs_load_fastq(f1, f2, 100.0)
        whole.count_kmers()
        pw, nw = whole.counts_buffer()
        engs, shards = [], []
        for r in range(world):
            g = Engine(k, e)
            shards.append(g.index_load_shard(index, r, world))
            g.counts_merge(pw, 0, nw)                      # the complete (already merged) count table
            if r == world - 1:
                g.sampling_init(100.0)
                g.pairs_load_fastq(f1, f2, 100.0)          # one rank also carries the reads, to vote
            engs.append(g)
        assert sum(s[0] for s in shards) == nc_all and sum(s[1] for s in shards) == nb_all
        if world <= nc_all:
            assert all(s[0] > 0 for s in shards)
        n_total, news = _sharded_scan_sequential(engs, case.hit_ratio, case.match_ratio, case.max_peak)
        n_whole = whole.ref_scan(case.hit_ratio, case.match_ratio, case.max_peak)
        assert n_total == n_whole == meta["raw_peaks"]
        loci_w, _ = whole.peaks_export(n_whole)
        pk_w = whole.peak_kmer_export(0, min(1 << k, 1 << 26))
        for g in engs:
            assert (g.peaks_export(n_total)[0] == loci_w).all()
            assert (g.peak_kmer_export(0, min(1 << k, 1 << 26)) == pk_w).all()
        voter = engs[-1]
        voter.vote()
        out = str(tmp_path / "interval.txt")
        voter.write_intervals(out)
        assert open(out).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()
        for g in engs:
            g.close()


def test_exchange_sharded_scan_world1_nccl(case_inputs, tmp_path):
    """Exchange.sharded_scan over nccl with one rank == lhgt_ref_scan"""
    from localhgt_amd.dist import Exchange
    from localhgt_amd.engine import Engine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    ex = Exchange.from_env(backend="nccl")
    name = "k24_base"
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    try:
        with Engine(case.k, case.e) as eng:
            eng.rng_seed(case.seed)
            eng.coder_generate()
            index = f"{fa2}.k{case.k}.h{case.e}.index.dat"
            eng.index_build(fa2, index, fa2 + ".genome.len.txt")
            eng.index_load_shard(index, 0, 1)
            eng.sampling_init(100.0)
            eng.pairs_load_fastq(f1, f2, 100.0)
            eng.count_kmers()
            assert ex.sharded_scan(eng, case.hit_ratio, case.match_ratio, case.max_peak) == meta["raw_peaks"]
            eng.vote()
            out = str(tmp_path / "interval.txt")
            eng.write_intervals(out)
            assert open(out).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()
    finally:
        ex.close()


@pytest.mark.parametrize("world,packed", [(2, False), (3, False), (8, False), (3, True), (8, True)])
def test_synthetic_reference_shards_scan_like_the_whole(world, packed):
    """bench.py's sharded form: every 'rank' generates only its contig range of the synthetic reference on the device
    (lhgt_synth_reference_shard); the exchanged scan must give the peak tables of one engine holding the whole reference.
    packed: the shards resident as packed bases (bench.py --ref-form packed at N > 1), the whole one in the index form"""
    from localhgt_amd.engine import Engine
    k, e, n_contigs, contig_len = 26, 3, 21, 60_000
    with Engine(k, e) as whole:
        whole.rng_seed(9)
        whole.coder_generate()
        coder = whole.coder_get()
        whole.synth_reference(7, n_contigs, contig_len)
        whole.synth_pairs(7, 8, n_contigs, contig_len, 0, 50_000)
        whole.count_kmers()
        pw, nw = whole.counts_buffer()
        engs = []
        for r in range(world):
            g = Engine(k, e)
            g.coder_set(coder)
            g.set_reference_form(packed)
            g.synth_reference_shard(7, n_contigs, contig_len, r, world)
            g.counts_merge(pw, 0, nw)
            engs.append(g)
        engs[-1].synth_pairs(7, 8, n_contigs, contig_len, 0, 50_000)     # one rank also carries the reads, to vote
        n_total, news = _sharded_scan_sequential(engs, 0.1, 0.08, 10**7)
        n_whole = whole.ref_scan(0.1, 0.08, 10**7)
        assert n_total == n_whole > 20 and sum(news) == n_total
        loci_w, _ = whole.peaks_export(n_whole)
        pk_w = whole.peak_kmer_export(0, 1 << k)
        for g in engs:
            assert (g.peaks_export(n_total)[0] == loci_w).all()
            assert (g.peak_kmer_export(0, 1 << k) == pk_w).all()
        whole.vote()
        engs[-1].vote()
        assert (engs[-1].peaks_export(n_total)[1] == whole.peaks_export(n_whole)[1]).all()
        for g in engs:
            g.close()
