"""The reference-SHARDED form of phase B (BASELINE configs[4]: the index sharded over 8 GPUs) at the addresses of the
configs it is named for, on the one GPU of the test box: the 8 shards live one after the other, their records (lhgt_ref_scan_emit)
are gathered by hand as localhgt_amd/dist.py: sharded_scan gathers them, and one engine replays them (lhgt_peaks_install).
Against the scan of the WHOLE reference on one engine -- loci, the 2^32-entry peak_kmer, the per-position flags (whose digest is
numbered over all indexed contigs, so the shards' digests add up) and the votes of the same pairs:
  * 50 Gbase, packed form (configs[4]; flat positions up to 5 x 10^10, plane words past 2^31);
  * 13 Gbase, the shards in the INDEX form (index words past 2^32 inside a shard), the whole one packed, under the -t 10
    emulation: the contig groups of split_ref cut across the shards (lhgt_ref_scan_group_counts / lhgt_set_group_totals).
Round 4 had the exchange kernels (emit_peaks, replay_regs) on 21 x 60 kb contigs only (tests/test_gpu_dist.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CL, E, K = 1_000_000, 3, 32
FLAG_BITS = 0b1111100
M64 = (1 << 64) - 1


def _shards_one_after_the_other(make_shard, world, counts, hit, match, max_peak, emu_threads, pairs):
    """-> (engine holding the installed peak tables and `pairs`, total new peaks, [acc, nz] of the shards' flag digests)"""
    import torch
    from localhgt_amd.dist import device_tensor
    pw, nw = counts
    news, n_sel, groups = [], 0, []
    if emu_threads > 1:                       # the threads' id ranges need the per-group totals before any record can be numbered
        for r in range(world):
            with make_shard(r) as g:
                g.counts_merge(pw, 0, nw)
                n, s = g.ref_scan_local(hit, match)
                groups.append(g.ref_scan_group_counts())
    totals = [sum(g[j] for g in groups) for j in range(emu_threads)] if emu_threads > 1 else None
    loci_parts, regs_parts, flag_sum, last, first_id = [], [], [0, 0], None, 0
    for r in range(world):
        g = make_shard(r)
        g.counts_merge(pw, 0, nw)
        n, s = g.ref_scan_local(hit, match)
        news.append(n)
        n_sel += s
        d = g.digest(g.DIGEST_FLAGS, FLAG_BITS)
        flag_sum = [(flag_sum[0] + d[0]) & M64, flag_sum[1] + d[1]]
        if totals is not None:
            first_id = g.set_group_totals(totals, max_peak)
        pl, pr, n_regs = g.ref_scan_emit(first_id + sum(news[:r]))
        loci_parts.append(device_tensor(pl, 8 * n, 0).view(torch.int32).clone() if n else torch.empty(0, dtype=torch.int32, device="cuda:0"))
        regs_parts.append(device_tensor(pr, 8 * n_regs, 0).view(torch.int32).clone() if n_regs else torch.empty(0, dtype=torch.int32, device="cuda:0"))
        torch.cuda.synchronize()
        if r < world - 1:
            g.close()
        else:
            last = g
    n_total = sum(news)
    loci_all, regs_all = torch.cat(loci_parts), torch.cat(regs_parts)
    if first_id:                              # no peak holds id 0: the loci table starts with an empty row
        loci_all = torch.cat([torch.zeros(2 * first_id, dtype=loci_all.dtype, device=loci_all.device), loci_all])
    torch.cuda.synchronize()
    last.peaks_install(n_total + first_id, n_sel, max_peak + first_id, loci_all.data_ptr(), regs_all.data_ptr(), regs_all.numel() // 2)
    pairs(last)
    return last, n_total, flag_sum


def _compare(whole, last, n_whole, n_total, flag_sum):
    assert n_total == n_whole > 1000
    for what in (whole.DIGEST_LOCI, whole.DIGEST_PEAK_KMER):
        assert last.digest(what) == whole.digest(what), what
    assert tuple(flag_sum) == tuple(whole.digest(whole.DIGEST_FLAGS, FLAG_BITS))
    whole.vote()
    last.vote()
    votes = whole.digest(whole.DIGEST_VOTES)
    assert last.digest(last.DIGEST_VOTES) == votes and votes[1] >= 1


def test_eight_shards_of_a_50_gbase_reference_scan_like_the_whole():
    from localhgt_amd.engine import Engine
    nc, world, n_pairs = 50_000, 8, 10_000_000
    with Engine(K, E) as whole:
        whole.rng_seed(1)
        whole.coder_generate()
        coder = whole.coder_get()
        whole.set_reference_form(True)
        whole.synth_reference(1, nc, CL)

        def pairs(eng):
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, nc, CL, 0, n_pairs)
            eng.synth_options(0, 20, 0)

        pairs(whole)
        whole.count_kmers()

        def make_shard(r):
            g = Engine(K, E)
            g.coder_set(coder)
            g.set_reference_form(True)
            g.synth_reference_shard(1, nc, CL, r, world)
            return g

        last, n_total, flag_sum = _shards_one_after_the_other(make_shard, world, whole.counts_buffer(), 0.1, 0.08, 300_000_000, 1, pairs)
        try:
            n_whole = whole.ref_scan(0.1, 0.08, 300_000_000)
            _compare(whole, last, n_whole, n_total, flag_sum)
        finally:
            last.close()


def test_eight_index_shards_of_13_gbase_under_thread_emulation_scan_like_the_whole():
    from localhgt_amd.engine import Engine
    nc, world, n_pairs, threads = 13_000, 8, 10_000_000, 10
    with Engine(K, E) as whole:
        whole.rng_seed(1)
        whole.coder_generate()
        coder = whole.coder_get()
        whole.set_reference_form(True)          # the whole reference as packed bases: the forms must agree as well
        whole.set_thread_emulation(threads)
        whole.synth_reference(1, nc, CL)

        def pairs(eng):
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, nc, CL, 0, n_pairs)
            eng.synth_options(0, 20, 0)

        pairs(whole)
        whole.count_kmers()

        def make_shard(r):
            g = Engine(K, E)
            g.coder_set(coder)
            g.set_thread_emulation(threads)
            g.synth_reference_shard(1, nc, CL, r, world)       # index form: 12 bytes per base, 19.5 GB per shard
            assert g.reference_info()["form"] == "index"
            return g

        last, n_total, flag_sum = _shards_one_after_the_other(make_shard, world, whole.counts_buffer(), 0.1, 0.08, 300_000_000, threads, pairs)
        try:
            n_whole = whole.ref_scan(0.1, 0.08, 300_000_000)
            _compare(whole, last, n_whole, n_total, flag_sum)
            # the threads' id ranges and sentinel lines: the interval files are the same text
            import os
            import tempfile
            with tempfile.TemporaryDirectory(prefix="lhgt_shard_") as d:
                a, b = os.path.join(d, "whole.txt"), os.path.join(d, "shards.txt")
                whole.write_intervals(a)
                last.write_intervals(b)
                assert open(a).read() == open(b).read() and open(a).read().count("\n") >= threads
        finally:
            last.close()
