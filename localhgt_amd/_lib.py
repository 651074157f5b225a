"""ctypes binding of liblocalhgt_hip.so (include/localhgt_hip.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible the product
path raises.  `load(require_gpu=False)` is only for symbol checks on a CPU-only box."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblocalhgt_hip.so")

CODER_SLOTS = 300
MAX_RANDOM = 50_000_000

ERRORS = {0: "OK", 1: "E_ARG", 2: "E_IO", 3: "E_HIP", 4: "E_FORMAT", 5: "E_STATE", 6: "E_TOO_MANY_PEAKS",
          7: "E_NOMEM", 8: "E_NO_DEVICE", 9: "E_EMULATION"}

_vp, _i, _l, _d, _f = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_float
_u8p, _u16p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_i32p, _lp, _fp, _dp = C.POINTER(C.c_int32), C.POINTER(C.c_long), C.POINTER(C.c_float), C.POINTER(C.c_double)
_cs = C.c_char_p

# name -> argtypes; every entry point of include/localhgt_hip.h (restype int unless noted)
SIGNATURES = {
    "lhgt_abi_version": [],
    "lhgt_last_error": [],
    "lhgt_device_count": [C.POINTER(C.c_int)],
    "lhgt_ctx_create": [_i, _i, _i, C.POINTER(_vp)],
    "lhgt_ctx_destroy": [_vp],
    "lhgt_rng_seed": [_vp, C.c_uint],
    "lhgt_coder_generate": [_vp],
    "lhgt_coder_set": [_vp, _u16p],
    "lhgt_coder_get": [_vp, _u16p],
    "lhgt_sampling_init": [_vp, _d],
    "lhgt_sampling_get": [_vp, _fp, _l],
    "lhgt_sampling_reserve": [_vp, _l],
    "lhgt_sampling_begin": [_vp],
    "lhgt_pool_trim": [],
    "lhgt_hash_sequence": [_vp, _cs, _l, _u32p, _u8p],
    "lhgt_index_build": [_vp, _cs, _cs, _cs, _lp, _lp],
    "lhgt_index_load": [_vp, _cs, _lp, _lp],
    "lhgt_index_load_shard": [_vp, _cs, _i, _i, _lp, _lp],
    "lhgt_index_from_memory": [_vp, _u8p, _u64p, _l],
    "lhgt_set_reference_form": [_vp, _i],
    "lhgt_reference_info": [_vp, C.POINTER(C.c_int), _u64p],
    "lhgt_index_read_coder": [_vp, _cs],
    "lhgt_reference_load_fasta": [_vp, _cs, _cs, _lp, _lp],
    "lhgt_fasta_scan": [_cs, _i, _cs, _lp, _lp, _lp],
    "lhgt_fastq_sam_ratio": [_cs, _d, _dp, _lp],
    "lhgt_pairs_load_fastq": [_vp, _cs, _cs, _d, _i, _i, _l, _lp, _lp],
    "lhgt_fastq_parse_digest": [_cs, _cs, _d, _fp, _i, _i, _l, _i, _l, _lp, _lp, _u64p],
    "lhgt_fastq_plan_chunk_bytes": [],
    "lhgt_fastq_plan_part": [_cs, _l, _i, _i, _u64p, _lp, _l, _lp, _lp, _lp],
    "lhgt_pairs_load_fastq_planned": [_vp, _cs, _cs, _d, _u64p, _lp, _l, _u64p, _lp, _l, _i, _i, _lp, _lp],
    "lhgt_fastq_parse_digest_planned": [_cs, _cs, _d, _fp, _i, _i, _l, _i, _l, _i, _u64p, _lp, _l, _u64p, _lp, _l, _i, _i, _i, _lp, _lp,
                                        _u64p, _lp],
    "lhgt_set_thread_emulation": [_vp, _i],
    "lhgt_fastq_thread_chunks": [_cs, _l, _i, _lp, _lp, _lp],
    "lhgt_fastq_thread_entry": [_cs, _l, _l],
    "lhgt_ingest_last_path": [_cs, _l],
    "lhgt_fastq_pair_chunk_bytes": [_cs, _cs, _lp, _lp],
    "lhgt_fastq_parse_rate": [_cs, _cs, _d, _fp, _i, _l, _i, _u64p, _lp, _l, _u64p, _lp, _l, _i, _i, _lp, _lp, _lp, _dp, _u64p],
    "lhgt_fastq_parse_digest_threads": [_cs, _cs, _d, _fp, _i, _i, _l, _i, _l, _i, _lp, _lp, _u64p, _lp],
    "lhgt_pairs_append": [_vp, _u8p, _u64p, _u8p, _u64p, _l, _u8p],
    "lhgt_pairs_append_flags": [_vp, _u8p, _u64p, _u8p, _u64p, _l, _u8p],
    "lhgt_pairs_batches": [_vp, _lp],
    "lhgt_pairs_batch_info": [_vp, _l, _lp, _u64p, C.POINTER(C.c_int)],
    "lhgt_pairs_store_write": [_vp, _cs, C.c_uint64, _l, _lp, _lp, _u64p],
    "lhgt_fastq_pack_host": [_cs, _cs, _cs, C.c_uint64, _i, _lp, _lp, _lp, _u64p, C.POINTER(C.c_int)],
    "lhgt_pairs_load_packed": [_vp, _cs, C.c_uint64, _l, _l, _l, _d, _i, _lp, _lp, _lp, _lp, _i, _i, _lp, _lp],
    "lhgt_vote_groups_export": [_vp, _u32p, _i, C.POINTER(C.c_int)],
    "lhgt_registry_info": [_vp, C.POINTER(C.c_int), _u64p, _u64p],
    "lhgt_packed_read_rate": [_cs, C.c_uint64, _l, _l, _i, _i, _i, C.POINTER(C.c_double)],
    "lhgt_set_count_on_load": [_vp, _i],
    "lhgt_pairs_clear": [_vp],
    "lhgt_pairs_count": [_vp, _lp],
    "lhgt_count_kmers": [_vp],
    "lhgt_set_count_mode": [_vp, _i],
    "lhgt_counts_clear": [_vp],
    "lhgt_set_count_compat": [_vp, _i],
    "lhgt_coder_generate_count_diff": [_vp],
    "lhgt_reads_load_count_diff": [_vp, _cs, _l, _i, C.c_uint, _lp],
    "lhgt_counts_buffer": [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)],
    "lhgt_counts_merge": [_vp, _vp, C.c_size_t, C.c_size_t],
    "lhgt_filter_buffer": [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)],
    "lhgt_ref_scan": [_vp, _f, _f, _l, _lp],
    "lhgt_ref_scan_local": [_vp, _f, _f, _lp, _lp],
    "lhgt_ref_scan_group_counts": [_vp, _lp, _i],
    "lhgt_set_group_totals": [_vp, _lp, _i, _l, _lp],
    "lhgt_ref_scan_emit": [_vp, _l, C.POINTER(_vp), C.POINTER(_vp), _lp],
    "lhgt_peaks_install": [_vp, _l, _l, _l, _vp, _vp, _l],
    "lhgt_vote": [_vp],
    "lhgt_write_intervals": [_vp, _cs, _lp],
    "lhgt_faidx_build": [_cs, _cs, _lp],
    "lhgt_faidx_extract": [_cs, _cs, _cs, _i, _lp, _lp],
    "lhgt_counts_export_u8": [_vp, C.c_uint64, C.c_uint64, _u8p],
    "lhgt_counts_histogram": [_vp, _u64p],
    "lhgt_flags_export": [_vp, C.c_uint64, C.c_uint64, _u8p],
    "lhgt_peaks_export": [_vp, _i32p, _u8p, _l],
    "lhgt_peak_kmer_export": [_vp, C.c_uint64, C.c_uint64, _u32p],
    "lhgt_digest": [_vp, _i, C.c_uint64, _u64p],
    "lhgt_synth_reference": [_vp, C.c_uint64, _l, _l, _u8p],
    "lhgt_synth_reference_shard": [_vp, C.c_uint64, _l, _l, _i, _i, _u8p],
    "lhgt_synth_reference_cuts": [_vp, C.c_uint64, _l, _l, _u64p, _l, _u8p],
    "lhgt_synth_pairs": [_vp, C.c_uint64, C.c_uint64, _l, _l, _l, _l, _i, _u8p, _u8p],
    "lhgt_synth_options": [_vp, _i, _i, _l],
    "lhgt_synth_read_mix": [_vp, _i, _i],
    "lhgt_set_debug": [_vp, _i],
    "lhgt_set_cu_mask": [_vp, _u32p, _i],
    "lhgt_phase_ms": [_vp, _i, _fp],
    "lhgt_scan_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double), _lp, _lp],
    "lhgt_slot_list": [_vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "lhgt_slot_list_build_ms": [_vp, _dp],
    "lhgt_work_stats": [_vp, _i, _u64p],
    "lhgt_vote_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "lhgt_stream": [_vp, C.POINTER(_vp)],
    "lhgt_synchronize": [_vp],
}


class LocalHGTError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"liblocalhgt_hip: {ERRORS.get(code, code)}: {message}")
        self.code = code


_lib = None


def _bind_one_hip_runtime():
    """One HIP runtime per process.  torch ships its own libamdhip64.so.7 (and HSA runtime) next to libtorch; this library is
    linked against /opt/rocm's.  Same SONAME, different builds: whichever is loaded first serves both, and when /opt/rocm's
    comes first torch later reports "No HIP GPUs are available".  So if torch is installed and not loaded yet, load ITS runtime
    before ours (no `import torch`: single-GPU runs never need it); when torch is already in, ours binds to it by SONAME."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def load(require_gpu: bool = True):
    """Load the shared library and bind every declared symbol. Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). localhgt_amd has no CPU fallback.")
        _bind_one_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header and library out of sync
            fn.argtypes = argtypes
            fn.restype = C.c_char_p if name == "lhgt_last_error" else C.c_long if name in ("lhgt_fastq_plan_chunk_bytes", "lhgt_fastq_thread_entry") else C.c_int
        _lib = lib
    if require_gpu:
        n = C.c_int(0)
        _lib.lhgt_device_count(C.byref(n))
        if n.value < 1:
            raise RuntimeError("liblocalhgt_hip: no HIP device visible; localhgt_amd has no CPU fallback")
    return _lib


def gpu_available() -> bool:
    """whether a HIP device is visible (the packer's choice between the GPU and the host's CPUs; the compute path has no such choice)"""
    lib = load(require_gpu=False)
    n = C.c_int(0)
    lib.lhgt_device_count(C.byref(n))
    return n.value >= 1


def check(rc: int):
    if rc != 0:
        raise LocalHGTError(rc, (_lib.lhgt_last_error() or b"").decode(errors="replace"))
