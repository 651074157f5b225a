"""Did the run find what was planted?  The breakpoints of the synthetic sample (k_synth.hip: transfer_sites) and the reference's
own quality measure for this stage (paper_results/evaluation.py:64-76)."""
_M64 = (1 << 64) - 1


def _mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def planted_breakpoints(n_contigs, contig_len, sample_contigs=0, ref_seed=1, transfer_len=3000):
    """(1-based contig number, position) of every breakpoint the synthetic sample carries: sample genome pair i = recipient
    contig 2i with a 3 kb insert at r0, donor contig 2i+1 that lost [d0, d0 + 3 kb)"""
    n_sample = (sample_contigs & ~1) if 0 < sample_contigs <= n_contigs else (n_contigs // 2) & ~1
    span = contig_len - 3 * transfer_len
    out = []
    for i in range(n_sample // 2):
        h = _mix64((ref_seed * 0x51ED2701 + i) & _M64)
        r0, d0 = transfer_len + h % span, transfer_len + _mix64(h) % span
        out += [(2 * i + 1, r0), (2 * i + 2, d0), (2 * i + 2, d0 + transfer_len)]
    return out


def interval_recall(interval_path, breakpoints):
    """the fraction of true breakpoints that fall inside an extracted interval"""
    by_contig = {}
    with open(interval_path) as f:
        for ln in f:
            c, a, b = (int(x) for x in ln.split())
            by_contig.setdefault(c, []).append((a, b))
    hit = sum(1 for c, p in breakpoints if any(a <= p <= b for a, b in by_contig.get(c, ())))
    return {"breakpoints": len(breakpoints), "inside_an_interval": hit, "recall": round(hit / max(1, len(breakpoints)), 4),
            "interval_lines": sum(len(v) for v in by_contig.values())}
