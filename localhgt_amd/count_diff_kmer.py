"""`count_diff_kmer fq1 fq2 k ratio` -- phase A alone plus the occupancy of the count table
(reference: /root/reference/src/count_diff_kmer.cpp, driven by paper_results/count_table_empty_with_k.py:31).

Output lines are the reference's (C:45-48, 363):
    ###kmer_is <k> sample_ratio_is <ratio>
    <2^k>\\t<#slots != 3>\\t<#slots == 0>
    ####<2^k>\\t<weak_rate>\\t<empty_rate>
with e fixed at 3 (C:20).  The reference tool seeds its coder and its sampling from time(0)
(C:87-89, 223-225) and cannot be reproduced run to run; here `--seed` (default 1) fixes both, the
sampling rule is the engine's (`random_array[n % 5e7] < ratio`), and a k-mer with a non-ACGT base is
skipped as in `extract_ref` (the tool's `bool` coder collapses N to 1, C:155-160).

`--compat` gives the reference tool's own numbers instead: what its binary prints when time() returns `--compat-time`
(default 1) and its 10 threads run in creation order -- its one-draw-per-position coder, the bool coder that never rejects a
base, the '@'-scan chunk starts, `>>` token reads, the byte budget that makes chunks overlap, the per-chunk read length and the
per-chunk rand() % 100 sampling (C:53-153, 155-160, 216-238; the tests pin it against the reference binary run with those
two determinism shims, tests/golden/count_diff_kmer.json)."""
from __future__ import annotations

import argparse
import sys

import numpy as np

from .engine import Engine


def fmt_g(x: float) -> str:
    return f"{x:g}"  # std::cout default: 6 significant digits


def run(fq1: str, fq2: str, k: int, ratio: float, seed: int = 1, device: int = 0, out=sys.stdout, compat: bool = False, compat_time: int = 1):
    with Engine(k, 3, device) as eng:
        if compat:
            import os
            ratio = int(ratio)                       # `int down_sam_ratio = stod(sample_ratio)` (C:313)
            eng.rng_seed(compat_time)                # srand(time(0)) (C:223-225)
            eng.coder_generate_count_diff()
            eng.set_count_compat(True)
            size1 = os.path.getsize(fq1)             # both files are cut by fq1's size (C:328-355)
            eng.reads_load_count_diff(fq1, size1, ratio, compat_time)
            eng.reads_load_count_diff(fq2, size1, ratio, compat_time)
        else:
            eng.rng_seed(seed)
            eng.coder_generate()
            eng.sampling_init(float(ratio))
            eng.pairs_load_fastq(fq1, fq2, float(ratio))
        eng.count_kmers()
        hist = eng.counts_histogram().astype(np.int64)
    size = 1 << k
    empty, weak = int(hist[0]), int(size - hist[3])
    print(f"###kmer_is {k} sample_ratio_is {fmt_g(ratio)}", file=out)
    print(f"{size}\t{fmt_g(float(weak))}\t{fmt_g(float(empty))}", file=out)
    print(f"####{size}\t{fmt_g(float(np.float32(weak / size)))}\t{fmt_g(float(np.float32(empty / size)))}", file=out)
    return hist


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="count_diff_kmer")
    ap.add_argument("fq1")
    ap.add_argument("fq2")
    ap.add_argument("k", type=float)
    ap.add_argument("ratio", type=float, help="down-sampling ratio in percent (1-100)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--compat", action="store_true", help="reproduce the reference tool's own chunking, tokenising, coder and sampling quirks")
    ap.add_argument("--compat-time", type=int, default=1, help="the value of time(0) the reference tool would have seeded with")
    a = ap.parse_args(argv)
    run(a.fq1, a.fq2, int(a.k), a.ratio, a.seed, compat=a.compat, compat_time=a.compat_time)
    return 0


if __name__ == "__main__":
    sys.exit(main())
