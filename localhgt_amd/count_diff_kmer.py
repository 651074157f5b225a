"""`count_diff_kmer fq1 fq2 k ratio` -- phase A alone plus the occupancy of the count table
(reference: /root/reference/src/count_diff_kmer.cpp, driven by paper_results/count_table_empty_with_k.py:31).

Output lines are the reference's (C:45-48, 363):
    ###kmer_is <k> sample_ratio_is <ratio>
    <2^k>\\t<#slots != 3>\\t<#slots == 0>
    ####<2^k>\\t<weak_rate>\\t<empty_rate>
with e fixed at 3 (C:20).  The reference tool seeds its coder and its sampling from time(0)
(C:87-89, 223-225) and cannot be reproduced run to run; here `--seed` (default 1) fixes both, the
sampling rule is the engine's (`random_array[n % 5e7] < ratio`), and a k-mer with a non-ACGT base is
skipped as in `extract_ref` (the tool's `bool` coder collapses N to 1, C:155-160)."""
from __future__ import annotations

import argparse
import sys

import numpy as np

from .engine import Engine


def fmt_g(x: float) -> str:
    return f"{x:g}"  # std::cout default: 6 significant digits


def run(fq1: str, fq2: str, k: int, ratio: float, seed: int = 1, device: int = 0, out=sys.stdout):
    with Engine(k, 3, device) as eng:
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
    a = ap.parse_args(argv)
    run(a.fq1, a.fq2, int(a.k), a.ratio, a.seed)
    return 0


if __name__ == "__main__":
    sys.exit(main())
