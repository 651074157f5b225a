"""interval.txt + <ref>.genome.len.txt -> interval.txt.bed (reference: scripts/get_bed_file.py).

Same output bytes as the reference script: `name:start-end` per interval, start clamped to 1
(G:15-16), intervals shorter than 50 dropped (G:17-18), and one stdout line
`extracted ref length is: N` (G:62) that pipeline.sh redirects into `${sample}.log`.

Quirk Q7: when a contig of length <= k precedes an indexed one, the ids in
genome.len.txt and in the interval file disagree; the reference then dies with KeyError after a
partial .bed or silently mis-names contigs.  Here that inconsistency is reported instead."""
from __future__ import annotations

import sys


class InconsistentReferenceIds(RuntimeError):
    pass


def index2name(reffile: str):
    names = {}
    order = []
    for line in open(reffile + ".genome.len.txt"):
        arr = line.strip().split()
        names[int(arr[1])] = arr[0]       # G:46-53
        order.append(int(arr[1]))
    return names, order


def write_bed(reffile: str, interval_file: str, strict: bool = True) -> int:
    names, order = index2name(reffile)
    if strict and order != list(range(1, len(order) + 1)):
        raise InconsistentReferenceIds(
            f"{reffile}.genome.len.txt numbers contigs {order[:6]}...: a contig shorter than k precedes an indexed "
            "one, so interval ids (sequential) and these ids disagree (reference quirk); remove contigs <= k.")
    extract_len = 0
    with open(interval_file) as f, open(interval_file + ".bed", "w") as h:
        for line in f:
            arr = line.strip().split()
            start, end = int(arr[1]), int(arr[2])
            s_start = arr[1]
            if start < 1:
                start, s_start = 1, "1"
            if abs(end - start) < 50:     # minimum fragment length
                continue
            print(f"{names[int(arr[0])]}:{s_start}-{arr[2]}", file=h)
            extract_len += end - start
    return extract_len


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    extract_len = write_bed(argv[0], argv[1])
    print("extracted ref length is:", extract_len)
    return 0


if __name__ == "__main__":
    sys.exit(main())
