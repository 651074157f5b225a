"""Rooflines of the three kernel families of a step against the 8 TB/s HBM peak -- and, since round 5, against the resource that
really bounds each of them (`bound`, `frac_of_bound`: BOUNDS below).

Two byte models per kernel, both per bench step:
  model   SURVEY.md 8d's algorithmic bytes -- one 64-B sector per probe of the REFERENCE's algorithm (714 probes per pair in
          phases A and C, e per reference base in phase B).  The kernels as built avoid most of those probes (radix partition,
          L2-resident bitmap, single-first / trio-first scan), so bytes_model / time can exceed the peak: `model_exceeded`.
  needed  the bytes the algorithm AS BUILT must move: every stream read or written once, every random probe that goes to HBM
          one 128-B line (tools/probe_shapes.hip: a random 4-byte load is a 128-B line fill on gfx950), probes answered on-chip
          (LDS fold, L2-resident bitmap, partition slices in LDS) nothing.  The probe and key counts are the run's own
          (lhgt_work_stats), so needed / time / peak <= 1 by construction up to counter noise: `frac_needed`; and
          `overfetch` = fabric bytes the counters saw / needed -- what the implementation moves beyond what its own algorithm asks.
"""
from .pmc import FETCH_SIZE_SCALE

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
LINE = 128                       # bytes a random probe moves (one fabric read request)
# request-rate ceilings of the memory system, measured by tools/probe_rates.hip (profiles/r01_probe_rates_microbench.txt,
# profiles/r02/probe_shapes_microbench.txt): random 4-byte loads that miss to HBM / that hit in L2
CEIL_HBM_GREQ, CEIL_L2_GREQ = 56.0, 254.0
HBM_CEILING = ("hbm_read_requests", "TCC_EA0_RDREQ_sum", CEIL_HBM_GREQ, "tools/probe_shapes.hip: random 4-byte loads from a 1 GiB table, 128-B line fills/s")
L2_CEILING = ("l2_read_requests", "TCP_TCC_READ_REQ_sum", CEIL_L2_GREQ, "tools/probe_rates.hip: random 4-byte loads from an L2-resident table")
# What BOUNDS each kernel (round 5).  The HBM fractions above stay as they were; next to them `bound` names the
# resource the kernel sits on and `frac_of_bound` how much of that resource's ceiling it uses:
#   hbm_lines    ref_flags*, the dense vote: random probes, every one a 128-B line fill.  Ceiling = the fabric's line rate, 56 G
#                fills/s = 7.2 TB/s (tools/probe_shapes.hip; the guide's HBM figures are for streams: 8 TB/s spec, 6.3 achievable)
#   l2_requests  the queued and the fold vote: 714 bitmap probes per pair answered by the XCDs' L2s.  Ceiling = the guide's L2
#                figure, 34.5 TB/s in 128-B requests = 269.5 G requests/s (MI355X_MICROARCH.md "L2 (per XCD)"); our own
#                microbenchmark of random 4-byte loads from an L2-resident table stays in request_rate (254 G/s)
#   lds_random   phase A: six random LDS operations per key over its three kernels (ticket + slot store, ticket + slot store, read +
#                compare-and-swap).  Ceiling = 6.5 lane-operations per clock and CU -- the guide's bank model (32 banks, a wave's 64
#                random lanes in two groups: 3.5 deep each) and what profiles/r04/phase_a_direct_ablation_final.txt measured -- x 256
#                CUs x 2.4 GHz = 4.0 T operations/s
L2_GUIDE_GREQ = 34.5e12 / LINE / 1e9
LDS_RANDOM_TOPS = 6.5 * 256 * 2.4e9 / 1e12
BOUNDS = {
    "hbm_lines": {"unit": "G line fills/s", "ceiling": CEIL_HBM_GREQ, "ceiling_source": "tools/probe_shapes.hip (profiles/r02/probe_shapes_microbench.txt): the fabric's random 128-B line rate = 7.2 TB/s"},
    "l2_requests": {"unit": "G requests/s", "ceiling": round(L2_GUIDE_GREQ, 1), "ceiling_source": "MI355X_MICROARCH.md, L2 (per XCD): 34.5 TB/s, in 128-B requests"},
    "lds_random": {"unit": "T lane-operations/s", "ceiling": round(LDS_RANDOM_TOPS, 2), "ceiling_source": "6.5 random lane-operations per clock and CU (guide's LDS bank model; profiles/r04/phase_a_direct_ablation_final.txt) x 256 CUs x 2.4 GHz"},
}


def model_bytes_per_pair(L, k, e):
    """one 64 B sector per probe + packed bases"""
    return 2 * (L - k + 1) * e * 64 + (2 * L + 3) // 4


def read_store_bytes(pairs, L):
    """what a read-side kernel streams in per step: per read the three bit-planes (ceil(L/32) + 1 words each) and its
    descriptor (u32 offset + u16 length)"""
    return pairs * 2 * (3 * ((L + 31) // 32 + 1) * 4 + 6)


def phase_a_chunks(pairs, direct):
    """launches of each of phase A's three kernels per step: a batch of the read store holds 16 Mi pairs and is counted in chunks of
    8 Mi pairs (round 4's direct form) or 4 Mi"""
    batch, chunk = 16 << 20, (8 << 20) if direct else (4 << 20)
    return (pairs // batch) * (batch // chunk) + -(-(pairs % batch) // chunk)


def is_direct(k, e, L):
    return (k, e) == (32, 3) and L <= 159


def needed_bytes(L, k, e, pairs, ref_bases, n_contigs, packed, stats, partitioned, scan_form, vote_form):
    """{family: (bytes per step, formula)} for the algorithm as built; stats = Engine.work_stats() of one step"""
    table = (1 << k) // 4                                   # 2-bit count table
    reads = read_store_bytes(pairs, L)
    keys = stats.get("count_keys") or pairs * 2 * (L - k + 1) * e
    out = {}
    if partitioned:
        direct = is_direct(k, e, L)                      # round 4's direct form: 24-bit level-1 keys, chunks of 8 Mi pairs
        n_chunks = phase_a_chunks(pairs, direct)
        bpk = 10 if direct else 12
        out["count_A"] = (reads + bpk * keys + n_chunks * 2 * table,
                          f"reads {reads} + {bpk} B x {keys} keys ({'3 written + 3 read' if direct else '4 written + 4 read'} + 2 written + 2 read over the three passes) + {n_chunks} chunks x 2 x {table} B of table slices in and out")
    else:
        out["count_A"] = (reads + 2 * table, f"reads {reads} + the cache-resident table in and out 2 x {table}")
    n_pos = max(0, ref_bases - n_contigs * (k - 1))          # positions with a k-mer
    probes = stats.get("scan_probes") or n_pos * e
    stream = ref_bases * 3 // 8 if packed else n_pos * 4 * e
    if scan_form == "slot-single":
        # flags preset and the not-a-base plane read (no_kmer_flags), the slot list streamed, one line of bases per position followed, the
        # probes of the followed positions and of the runs (32 positions of every 250: one line of bases per run)
        followed = stats.get("scan_followed", 0)
        runs = n_pos // 250
        out["ref_flags"] = (ref_bases + ref_bases // 8 + 6 * n_pos + (followed + probes + runs) * LINE,
                            f"flag bytes preset {ref_bases} + not-a-base plane {ref_bases // 8} + slot list 6 B x {n_pos} positions + {followed} positions followed "
                            f"({followed / max(1, n_pos):.3f} of them) x {LINE} B of bases + {probes} table probes x {LINE} B ({probes / max(1, n_pos):.3f} per position) "
                            f"+ {runs} runs x {LINE} B of bases")
    elif scan_form == "slot-first":
        # the slot list streamed (6 B per position with a k-mer), one line of hashes / bases per position followed, their probes, the
        # flags cleared and the trio positions written (counted with the clearing)
        followed = stats.get("scan_followed", 0)
        list_b = stats.get("slot_list_bytes") or 6 * n_pos
        with_mid = list_b > 8 * n_pos                       # the entries carry their second-largest hash: a followed position asks the table first,
        ref_lines = max(0, probes - followed) if with_mid and e > 2 else followed     # and only those that pass (= the second probes) read the reference
        out["ref_flags"] = (list_b + (ref_lines + probes) * LINE + ref_bases,
                            f"slot list {list_b} B ({list_b / max(1, n_pos):.0f} B x {n_pos} positions) + {followed} positions followed ({followed / max(1, n_pos):.3f} of them), "
                            f"{ref_lines} of them read {LINE} B of {'bases' if packed else 'stored hashes'} + {probes} table probes x {LINE} B + flag bytes cleared {ref_bases}")
    else:
        out["ref_flags"] = (probes * LINE + stream + 2 * ref_bases,
                            f"{probes} table probes x {LINE} B ({scan_form}: {probes / max(1, n_pos):.3f} per position) + "
                            f"{'packed planes' if packed else 'index words'} {stream} + flag and state bytes written 2 x {ref_bases}")
    if vote_form in ("queued", "fold"):
        hb = stats.get("vote_hbm_probes", 0)
        out["vote_kernel"] = (reads + hb * LINE,
                              f"reads {reads} + {hb} probes of peak_kmer x {LINE} B ({hb / max(1, pairs):.2f} per pair survive the "
                              f"{'LDS fold and the ' if vote_form == 'fold' else ''}L2-resident bitmap, which costs no HBM byte)")
    elif vote_form == "shared":
        # round 6: reads grouped by their smallest hash, a workgroup fetches every slot of the distinct k-mers of its <= 64 reads once; the
        # records are read by the grouping (once per store) and by the probe kernel, an 80-byte record per read is written and read by the
        # filter, and the pairs that can vote are probed again, every probe, by the generic kernel
        fetched = stats.get("vote_shared_fetches", 0) + stats.get("vote_shared_outside", 0)
        again = stats.get("vote_revoted_pairs", 0)
        probes_pp = 2 * (L - k + 1) * e
        out["vote_kernel"] = (reads + fetched * LINE + pairs * 2 * 2 * 80 + again * probes_pp * LINE,
                              f"reads {reads} + {fetched} slots of peak_kmer x {LINE} B ({fetched / max(1, pairs):.1f} line fills per pair for "
                              f"{keys / max(1, pairs):.0f} probes: overlapping reads share them) + per-read records written and read {pairs * 2 * 2 * 80} + "
                              f"{again} pairs that can vote ({100.0 * again / max(1, pairs):.1f} %) x {probes_pp} probes x {LINE} B in the generic kernel")
    else:
        out["vote_kernel"] = (reads + keys * LINE, f"reads {reads} + {keys} probes of peak_kmer x {LINE} B (no on-chip filter: dense peak set)")
    return out


def roofline_entry(kernel, desc, ms_step, launches, model_b, needed, traffic_rec, source, ceiling, bound="hbm_lines", lds_ops=None):
    """one kernel family against the HBM peak.  frac = frac_fabric = (2 x FETCH_SIZE + WRITE_SIZE) / time / peak (every fabric read
    request of these kernels is a 128-B line fill tallied at 64 B; an estimate of fabric bytes, Infinity-Cache hits included);
    frac_raw = the counters as they are; frac_model / frac_needed / overfetch: see the module header."""
    ent = {"kernel": kernel, "what": desc, "bound": bound, "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(ms_step, 3),
           "launches_per_step": launches, "launch_ms": round(ms_step / launches, 3) if launches else None,
           "bytes_model": model_b}
    s = ms_step * 1e-3
    if s <= 0:
        return ent
    fm = model_b / s / 1e9 / HBM_PEAK_GBS
    ent.update({"frac_model": round(fm, 4), "model_exceeded": bool(fm > 1.0)})
    ent["bound_ceiling"] = dict(BOUNDS[bound])
    if bound == "lds_random" and lds_ops:                  # from the run's own key count, no counter needed
        ent["bound_ceiling"]["value"] = round(lds_ops / s / 1e12, 3)
        ent["frac_of_bound"] = round(lds_ops / s / 1e12 / LDS_RANDOM_TOPS, 3)
    if needed:
        ent.update({"bytes_needed": int(needed[0]), "needed_is": needed[1], "frac_needed": round(needed[0] / s / 1e9 / HBM_PEAK_GBS, 4)})
    if traffic_rec:
        b = traffic_rec["bytes"]
        ach = b / s / 1e9
        ent.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": b // max(1, launches or 1),
                    "traffic_per_step": b, "fetch_size_scale": FETCH_SIZE_SCALE, "traffic_source": source})
        if needed and needed[0]:
            ent["overfetch"] = round(b / needed[0], 3)
        if traffic_rec.get("bytes_raw"):
            ent["frac_raw"] = round(traffic_rec["bytes_raw"] / s / 1e9 / HBM_PEAK_GBS, 4)
        req = traffic_rec.get(ceiling[0])
        if req:
            g = req / s / 1e9
            ent["request_rate"] = {"value": round(g, 1), "unit": "G requests/s", "counter": ceiling[1], "ceiling": ceiling[2],
                                   "frac_of_ceiling": round(g / ceiling[2], 3), "ceiling_source": ceiling[3]}
        for key in ("l2_hit_rate", "l2_hits", "l2_misses", "hbm_read_requests", "l2_read_requests"):
            if traffic_rec.get(key) is not None:
                ent[key] = traffic_rec[key]
        breq = traffic_rec.get({"hbm_lines": "hbm_read_requests", "l2_requests": "l2_read_requests"}.get(bound, ""))
        if breq:
            ent["bound_ceiling"]["value"] = round(breq / s / 1e9, 1)
            ent["frac_of_bound"] = round(breq / s / 1e9 / BOUNDS[bound]["ceiling"], 3)
    else:
        ent.update({"achieved": None, "frac": None, "traffic": None,
                    "traffic_source": "none: the rocprofv3 --pmc passes failed and profiles/traffic_per_launch.json was measured on other sources"})
    return ent


def vote_form_of(vote, stats):
    """which vote kernel lhgt_vote took: from Engine.vote_info(), else from what it counted"""
    if vote:
        return vote["form"] if vote["form"] in ("fold", "queued", "shared") else "dense"
    if stats.get("vote_l2_probes"):
        return "fold"
    if stats.get("vote_hbm_probes") or stats.get("vote_revoted_pairs"):
        return "queued"
    return "dense"


def rooflines(k, e, L, pairs, ref_bases, n_contigs, packed, per_ms, scan, n_peaks, traffic, src, stats, vote=None):
    """entries of the three kernel families of one workload (phase A's as one), and which one dominates the step.
    per_ms = phase_ms(0..3): A, B, C, the ref_flags kernel alone"""
    model_pairs = model_bytes_per_pair(L, k, e) * pairs
    model_ref = ref_bases * (64 * e) + (ref_bases // 4 if packed else ref_bases * 4 * e)   # 204 B per base / 0.25 + 192
    n_batches = -(-pairs // (16 << 20))
    direct = is_direct(k, e, L)
    n_chunks = phase_a_chunks(pairs, direct)
    partitioned = k >= 26
    vform = vote_form_of(vote, stats or {})
    need = needed_bytes(L, k, e, pairs, ref_bases, n_contigs, packed, stats or {}, partitioned, scan["form"], vform) if stats is not None else {}
    kern = {"count_A": per_ms[0], "ref_flags": per_ms[3], "vote_kernel": per_ms[2]}
    scan_kernel = {"single-first": "ref_flags_lite", "trio-first": "ref_flags_trio", "slot-first": "ref_flags_slots",
                   "slot-single": "no_kmer_flags+ref_single_slots+ref_trio_runs"}.get(scan["form"], "ref_flags")
    vote_kernel = {"fold": "vote_kernel_fold", "queued": "vote_kernel_queued", "shared": "vs_probe+vs_filter+vote_kernel"}.get(vform, "vote_kernel")
    info = {
        "count_A": (("part_reads_direct+part_keys16_direct+part_apply2" if direct else "part_scatter_reads+part_scatter_keys16+part_apply") if partitioned else "count_direct",
                    f"phase A kernel family, {n_chunks} chunks of <= {8 if direct else 4} Mi pairs per step: {2 * (L - k + 1) * e} table updates per pair",
                    model_pairs, 3 * n_chunks if partitioned else n_batches, HBM_CEILING, "lds_random" if partitioned else "hbm_lines"),
        "ref_flags": (scan_kernel, {"single-first": "phase B on a nearly saturated table: one probe per base until a hash reads 3, all e at every 8th base",
                                    "trio-first": "phase B on a sparse table: probes per base until a hash does not read 3",
                                    "slot-first": "phase B on a sparse table, the first probe of every position answered from the slot list of the resident reference",
                                    "slot-single": "phase B on a nearly saturated table: `single` of every position from the slot list of the resident reference (positions whose "
                                                   "smallest hash's slot reads 3 are not touched), the trio lower bound from runs of 32 positions"}.get(
                                        scan["form"], "phase B: e random 2-bit table probes per reference base") + "; 1 launch per step",
                      model_ref, 1, HBM_CEILING, "hbm_lines"),
        "vote_kernel": (vote_kernel, f"phase C read re-scan, {2 * (L - k + 1) * e} probes per pair "
                        + {"fold": "screened by a 128 KiB LDS fold, then the L2-resident bitmap, then peak_kmer",
                           "queued": "answered by the L2-resident bitmap except for its survivors",
                           "shared": "answered from LDS sets of the distinct slots of 32 reads that share a champion k-mer, each slot fetched once"}.get(vform, "into peak_kmer")
                        + f"; {1 if vform == 'shared' else n_batches} launches per step", model_pairs, 1 if vform == "shared" else n_batches,
                        HBM_CEILING if vform in ("dense", "shared") else L2_CEILING, "hbm_lines" if vform in ("dense", "shared") else "l2_requests"),
    }
    keys_a = (stats or {}).get("count_keys") or pairs * 2 * (L - k + 1) * e
    roof = {ph: roofline_entry(info[ph][0], info[ph][1], kern[ph], info[ph][3], info[ph][2], need.get(ph), traffic.get(ph) if src else None, src, info[ph][4],
                               bound=info[ph][5], lds_ops=6 * keys_a if ph == "count_A" else None)
            for ph in kern}
    dominant = max(kern, key=kern.get)                      # over A (as one entry), B's probe kernel and C
    return roof, dominant
