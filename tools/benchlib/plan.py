"""Per-GPU HBM plan of a bench workload, checked before anything is allocated: a run that cannot fit fails at once
with the sizes, not minutes later inside a hipMalloc."""
GB = 1e9
HBM_BYTES = 288 * GB           # MI355X


def memory_plan(pairs, ref_bases, n_contigs, k=32, e=3, L=150, packed=False, world=1, shard_index=False, slot_list=False):
    """bytes resident on ONE GPU during a step.  pairs = read pairs of this GPU's shard; the reference is whole on every GPU
    unless shard_index (then 1/world of it)."""
    share = 1.0 / world if shard_index else 1.0
    n_pos = ref_bases * share
    wpr = (L + 31) // 32 + 1
    direct = (k, e) == (32, 3) and L <= 159               # phase A's direct form: chunks of 8 Mi pairs, 24-bit level-1 keys
    chunk_pairs = min(pairs, (8 if direct else 4) << 20)
    keys = chunk_pairs * 2 * (L - k + 1) * e
    nb = 1 << max(0, k - 16)
    plan = {
        "reference": n_pos * 3 / 8 + 64 if packed else (n_pos - n_contigs * share * (k - 1)) * 4 * e + 4 * n_contigs * share,
        # 6 bytes per position in regions sized from a sampled histogram (+ 7.5 % at 13 Gbase), bucket starts and ends
        "slot_list": 6 * n_pos * 1.075 + 16 * ((1 << max(0, k - 14)) + 1) if slot_list and packed and n_pos < (1 << 34) else 0,
        "per_position_flags_and_state": 2 * n_pos,
        "tile_tables": (n_pos / 2000 + n_contigs * share) * (8 + 1 + 4 + 4) + n_contigs * share * 24,
        "read_store": pairs * 2 * (3 * wpr * 4 + 4 + 2) + pairs,
        "count_table": (1 << k) / 4,
        "count_table_saturation_bitmap": (1 << k) / 4 / 64 / 8,
        "peak_kmer": (1 << k) * 4,
        # two buffers of `need` x 4 and `need` x 2 bytes, need = keys + 1/16 + 512 per final bucket (the direct form's level-1 pieces
        # hold 24-bit keys and use three quarters of the first)
        "partition_key_buffers": (keys + keys / 16 + nb * 512 + 64) * (4 + 2) + (3 * 65536) * 4 if k >= 26 else 0,
        "vote_bitmap_fold_and_lists": (1 << 25) / 8 + 128 * 1024 + (min(pairs, 16 << 20) + 1) * 4,
        "exchange_buffers": 2 * (1 << k) / 4 if world > 1 else 0,     # all_to_all receive (world slices of 1/world) + the gathered slice's clone
        "synthetic_generator_staging": 0.5 * GB,
    }
    plan = {kk: int(v) for kk, v in plan.items()}
    plan["total"] = sum(plan.values())
    return plan


def check_fits(plan, capacity=HBM_BYTES, what=""):
    """raises SystemExit with the sized plan when it cannot fit (5 % of the device kept for the runtime and fragmentation)"""
    if plan["total"] > 0.95 * capacity:
        rows = ", ".join(f"{kk} {v / GB:.1f} GB" for kk, v in plan.items() if kk != "total" and v > 0.05 * GB)
        raise SystemExit(f"bench: {what} needs {plan['total'] / GB:.1f} GB per GPU, the device has {capacity / GB:.0f} GB: {rows}. "
                         f"Use --ref-form packed (3/8 byte per base instead of 12), fewer --pairs, or --shard-index at N > 1.")
    return plan
