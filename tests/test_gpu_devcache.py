"""The process-wide cache of large device blocks (localhgt_amd/csrc/cabi.hip: dev_free / dev_alloc_raw) under two contexts on one
GPU: a block one context parks while its kernels are still queued -- key buffers and read batches that regrow in
mid-stream -- may be handed to the other context only once the device has drained what was queued before the park.  Two host
threads, each with a context of its own, grow their batches step by step (every step frees and re-allocates blocks of 64 MiB and
more, of size classes the two contexts share) and count; every table must equal the one a lone context computes."""
import threading

import pytest

pytestmark = pytest.mark.gpu
K, E, NC, CL = 26, 3, 40, 100_000
STEPS = [60_000, 150_000, 90_000, 400_000, 250_000, 700_000]


def _table(eng, seed, n):
    eng.pairs_clear()
    eng.counts_clear()
    eng.synth_pairs(1, seed, NC, CL, 0, n)
    eng.set_count_mode(1)                  # the radix partition: key buffers sized by the batch
    eng.count_kmers()
    return eng.digest(eng.DIGEST_COUNTS)


def test_two_contexts_share_the_block_cache_while_regrowing():
    from localhgt_amd.engine import Engine, pool_trim
    pool_trim()
    with Engine(K, E) as lone:
        lone.rng_seed(1)
        lone.coder_generate()
        lone.synth_reference(1, NC, CL)
        want = {(seed, n): _table(lone, seed, n) for seed in (2, 3) for n in STEPS}
    got, errs = {}, []

    def worker(seed):
        try:
            for rnd in range(2):
                with Engine(K, E) as eng:      # closing parks the context's blocks for the other thread's next regrowth
                    eng.rng_seed(1)
                    eng.coder_generate()
                    eng.synth_reference(1, NC, CL)
                    for n in (STEPS if rnd == 0 else STEPS[::-1]):
                        got[(seed, n, rnd)] = _table(eng, seed, n)
        except Exception as ex:               # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=worker, args=(seed,)) for seed in (2, 3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for (seed, n, rnd), d in got.items():
        assert d == want[(seed, n)], (seed, n, rnd)
    pool_trim()
