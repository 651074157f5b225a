#!/usr/bin/env python3
"""Can phase B's probe kernel run beside phase A?  DESIGN.md 8 names the lever: a count-table slot that reads 3 is
final while A still counts (the counters only grow and saturate, E:1082-1084), so the single-first scan could settle positions
against the unfinished table.  Whatever that would save is bounded by what the probe kernel gets done WHILE A runs -- measured
here without building the two-pass scan: two contexts on one GPU, one looping phase A of configs[2] (100 M pairs), the other
looping its scan (B) on its own finished table, alone and side by side, without CU masks (A's persistent grids are one workgroup per
CU: under a mask that takes CUs away they run in two rounds, profiles/r03/cu_share_phase_times.txt).  usage: ab_overlap.py [pairs]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localhgt_amd.engine import Engine   # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
NC, CL, K, E = 13000, 1_000_000, 32, 3


def make(seed):
    g = Engine(K, E)
    g.rng_seed(1)
    g.coder_generate()
    g.set_reference_form(True)
    g.synth_reference(1, NC, CL)
    g.synth_pairs(1, seed, NC, CL, 0, pairs, 150)
    g.counts_clear()
    g.count_kmers()
    g.ref_scan(0.1, 0.08, 300_000_000)
    return g


ga, gb = make(2), make(2)
gb.pairs_clear()                      # the scanning context keeps its finished table, not its reads


def loop_a(n, out):
    for _ in range(n):
        ga.counts_clear()
        ga.count_kmers()
        out.append(ga.phase_ms(0))


def loop_b(n, out):
    for _ in range(n):
        gb.ref_scan(0.1, 0.08, 300_000_000)
        out.append((gb.phase_ms(1), gb.phase_ms(3)))


a_alone, b_alone = [], []
loop_a(3, a_alone)
loop_b(3, b_alone)
A, B, Bk = min(a_alone), min(x[0] for x in b_alone), min(x[1] for x in b_alone)
print(f"alone: phase A {A:.1f} ms; phase B {B:.1f} ms (its probe kernel {Bk:.1f}), {gb.scan_info()['form']}", flush=True)
for na, nb in ((4, 4), (4, 8)):
    ra, rb = [], []
    t0 = time.time()
    th = [threading.Thread(target=loop_a, args=(na, ra)), threading.Thread(target=loop_b, args=(nb, rb))]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = (time.time() - t0) * 1e3
    serial = na * A + nb * B
    print(f"side by side, {na} x A and {nb} x B: wall {wall:.0f} ms against {serial:.0f} ms one after the other (gain {serial / wall:.3f}); "
          f"A took {sum(ra) / len(ra):.1f} ms each (alone {A:.1f}), B {sum(x[0] for x in rb) / len(rb):.1f} (alone {B:.1f}), "
          f"its probe kernel {sum(x[1] for x in rb) / len(rb):.1f} (alone {Bk:.1f})", flush=True)
ga.close()
gb.close()
