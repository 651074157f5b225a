#!/usr/bin/env python3
"""Phase B on a sparse table, trio-first kernel against the slot list (round 5; k_scan.hip: ref_flags_slots), on the 13 Gbase reference
resident as packed bases: the deep focused sample (100 M pairs from 300 genomes), a 10 M-pair sample of them, and the CLI's default
down-sampled regime on a realistic sample (6.67 M pairs from 300 genomes).  Same digests asserted.  usage: slot_list_leg.py [index|packed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine   # noqa: E402

NC, CL, K, E = 13000, 1_000_000, 32, 3
form = sys.argv[1] if len(sys.argv) > 1 else "packed"
g = Engine(K, E)
g.rng_seed(1)
g.coder_generate()
g.set_reference_form(form == "packed")
t0 = time.time()
g.synth_reference(1, NC, CL)
print(f"reference ({form}): {g.reference_info()}, {time.time() - t0:.1f} s", flush=True)


def scan(dbg, n=3):
    g.set_debug(dbg)
    best = None
    for _ in range(n):
        t = time.time()
        npk = g.ref_scan(0.1, 0.08, 300_000_000)
        wall = (time.time() - t) * 1e3
        r = (g.phase_ms(1), g.phase_ms(3), wall)
        best = r if best is None or r[0] < best[0] else best
    info = g.scan_info()
    dig = (npk, g.digest(g.DIGEST_LOCI), g.digest(g.DIGEST_PEAK_KMER), g.digest(g.DIGEST_FLAGS, 0b1111100))
    g.set_debug(0)
    return best, info, dig


for pairs, contigs, tag in ((100_000_000, 300, "deep focused"), (10_000_000, 300, "focused, 10 M pairs"), (6_670_000, 300, "default down-sampled, 300 genomes")):
    g.pairs_clear()
    g.synth_options(0, 20, contigs)
    g.synth_pairs(1, 2, NC, CL, 0, pairs)
    g.synth_options(0, 20, 0)
    g.counts_clear()
    g.count_kmers()
    a = g.phase_ms(0)
    (b_t, k_t, w_t), info_t, dig_t = scan(16384)
    t = time.time()
    (b_s, k_s, w_s), info_s, dig_s = scan(1 << 24)
    first = time.time() - t
    assert info_s["form"] == "slot-first", info_s
    assert dig_t == dig_s, (dig_t, dig_s)
    g.vote()
    c = g.phase_ms(2)
    print(f"{tag}: {pairs} pairs, table {100 * info_t['frac_slots_at_3']:.1f} % at 3, {dig_t[0]} raw peaks; phase A {a:.1f} ms, C {c:.1f} ms; "
          f"phase B trio-first {b_t:.1f} ms (probe kernel {k_t:.1f}, wall {w_t:.0f}), tiles treated exactly {info_t['tiles_exact']}; "
          f"slot-first {b_s:.1f} ms (sweep kernel {k_s:.1f}, wall {w_s:.0f}), tiles {info_s['tiles_exact']}; three scans incl. any build {first:.2f} s; "
          f"step {a + b_t + c:.0f} -> {a + b_s + c:.0f} ms, {pairs / (a + b_t + c) / 1e3:.1f} -> {pairs / (a + b_s + c) / 1e3:.1f} M pairs/s; list {g.slot_list()}", flush=True)

# the headline's sample (half of the catalogue, a saturated table): the single-first kernel against its list form (slot-single:
# the list rebuilt under the smallest hash; packed form only)
if form == "packed":
    g.pairs_clear()
    g.synth_pairs(1, 2, NC, CL, 0, 100_000_000)
    g.counts_clear()
    g.count_kmers()
    a = g.phase_ms(0)
    (b_t, k_t, w_t), info_t, dig_t = scan(4096)
    t = time.time()
    (b_s, k_s, w_s), info_s, dig_s = scan(4096 | (1 << 24))
    first = time.time() - t
    assert info_s["form"] == "slot-single", info_s
    assert dig_t == dig_s, (dig_t, dig_s)
    g.vote()
    c = g.phase_ms(2)
    print(f"configs[2]'s sample: table {100 * info_t['frac_slots_at_3']:.1f} % at 3, {dig_t[0]} raw peaks; phase A {a:.1f} ms, C {c:.1f} ms; "
          f"phase B single-first {b_t:.1f} ms (probe kernel {k_t:.1f}), tiles treated exactly {info_t['tiles_exact']}; "
          f"slot-single {b_s:.1f} ms (its kernels {k_s:.1f}), tiles {info_s['tiles_exact']}; three scans incl. the rebuild {first:.2f} s; "
          f"step {a + b_t + c:.0f} -> {a + b_s + c:.0f} ms, {1e5 / (a + b_t + c):.1f} -> {1e5 / (a + b_s + c):.1f} M pairs/s; list {g.slot_list()}", flush=True)
g.close()
