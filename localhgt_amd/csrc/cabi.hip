// cabi.hip -- context life cycle and the small accessors of the C-ABI (include/localhgt_hip.h).
#include <cstring>
#include "lhgt_common.hpp"

using namespace lhgt;

namespace lhgt {
// Position-sensitive checksum of a device array for parity tests at sizes whose tables cannot go through the host:
// sum over i of mix(i, v[i] & mask) mod 2^64 (order of summation free, so one pass of atomics).  BYTES = element size.
__device__ __forceinline__ uint64_t digest_mix(uint64_t i, uint64_t v) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull + v * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <class T>
__global__ void __launch_bounds__(256) digest_kernel(const T* __restrict__ v, uint64_t n, uint64_t mask, uint64_t index_base, unsigned long long* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0, nz = 0;
    for (; i < n; i += stride) {
        const uint64_t x = (uint64_t)v[i] & mask;
        acc += digest_mix(index_base + i, x);
        nz += x != 0;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { acc += __shfl_xor(acc, d, 64); nz += __shfl_xor(nz, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, (unsigned long long)acc); atomicAdd(out + 1, (unsigned long long)nz); }
}
}  // namespace lhgt

#include <mutex>
#include <unordered_map>
namespace lhgt {
namespace {
// A parked block remembers the park "generation"; a device that has been synchronised since (g_dev_synced) has finished every
// kernel that could still have been reading or writing the block when its owner let go of it.  hipFree used to give that guarantee
// implicitly (it synchronises the device); the cache gives it when the block is handed out again -- a block parked by one
// context's regrowth in mid-stream (key buffers, d_revote, staging) must not reach another context's stream while the first
// one's queued kernels still use it (two contexts on one GPU: tools/benchlib/legs.py pipelined_samples).
struct DevBlock { int device; size_t bytes; void* p; unsigned long long gen; };
std::mutex g_dev_mu;
std::vector<DevBlock> g_dev_free;                                   // cached blocks, oldest first
std::unordered_map<void*, std::pair<int, size_t>> g_dev_live;       // blocks handed out by dev_alloc_raw: device, class size
std::unordered_map<int, unsigned long long> g_dev_synced;           // per device: every block parked at or before this generation is quiescent
unsigned long long g_dev_gen = 0;
size_t g_dev_cached = 0;
size_t g_dev_cap = 0;                                               // 0 = not decided yet (dev_cache_cap)
thread_local lhgt_ctx* t_entry_ctx = nullptr;                       // the context of the C-ABI call running on this thread (LHGT_DEVICE_ENTRY)
// what is kept at most: a context's tables, buffers and batches at configs[1]..[2] sizes (96 GB) -- but never more than a third of
// the device (LHGT_DEV_CACHE_GB overrides; ranks that share a GPU, torch and RCCL do not see parked blocks as reclaimable)
size_t dev_cache_cap() {
    if (g_dev_cap) return g_dev_cap;
    size_t cap = (size_t)96 << 30;
    if (const char* s = getenv("LHGT_DEV_CACHE_GB")) cap = (size_t)(atof(s) * (double)(1ull << 30));
    else {
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) == hipSuccess && t / 3 < cap) cap = t / 3;
        else (void)hipGetLastError();
        if (const char* w = getenv("WORLD_SIZE")) if (atoi(w) > 1) cap = std::min(cap, (size_t)24 << 30);   // ranks may share this GPU (gloo mode)
    }
    g_dev_cap = cap ? cap : 1;
    return g_dev_cap;
}
size_t size_class(size_t bytes) {                                   // next eighth-step of a power of two, steps of at most 256 MiB
    size_t top = (size_t)1 << 26;                                   // (a 156 GB index must not grow by 16 GB for the sake of reuse)
    while ((top << 1) <= bytes) top <<= 1;
    const size_t step = std::min(top >> 3, (size_t)256 << 20);
    return (bytes + step - 1) / step * step;
}
// last resort of an allocation that would fail: the optional slot list (78-130 GB) of the context whose call is running on this
// thread, when no scan is using it.  Other threads' contexts are left alone (their lists may be in use).
bool drop_optional(void) {
    lhgt_ctx* c = t_entry_ctx;
    if (c && c->d_rg_buf) {                                         // the registry by partition's record buffers: idle outside register_partitioned, sized again by the next scan
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        (void)dev_free(c->d_rg_buf);
        c->d_rg_buf = nullptr;
        c->rg_buf_bytes = 0;
        (void)big_release_all();
        return true;
    }
    if (!c || c->sl_state != 1 || c->sl_in_use) return false;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    slot_list_drop(c);
    c->sl_state = -1;                                               // not built again for this reference: the memory is wanted elsewhere
    (void)big_release_all();                                        // the list's blocks were parked by slot_list_drop
    if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] out of device memory: the slot list was dropped\n");
    return true;
}
hipError_t malloc_hard(void** p, size_t bytes) {
    // test hook (tests/test_gpu_devcache.py): LHGT_TEST_FAIL_ALLOC=<bytes> makes the first attempt of an allocation of that size or more
    // fail as if the device were full WHILE the calling context holds an idle slot list -- the path that drops the list and tries again
    static const long fail_from = getenv("LHGT_TEST_FAIL_ALLOC") ? atol(getenv("LHGT_TEST_FAIL_ALLOC")) : 0;
    const bool pretend = fail_from > 0 && bytes >= (size_t)fail_from && t_entry_ctx && t_entry_ctx->sl_state == 1 && !t_entry_ctx->sl_in_use;
    hipError_t e = pretend ? hipErrorOutOfMemory : hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && big_release_all()) { (void)hipGetLastError(); e = hipMalloc(p, bytes); }
    if (e == hipErrorOutOfMemory && drop_optional()) { (void)hipGetLastError(); e = hipMalloc(p, bytes); }
    return e;
}
}  // namespace
void entry_context(lhgt_ctx* ctx) { t_entry_ctx = ctx; }
hipError_t dev_alloc_raw(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes < DEV_CACHE_MIN) return malloc_hard(p, bytes);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const size_t cls = size_class(bytes);
    {
        std::unique_lock<std::mutex> lk(g_dev_mu);
        for (size_t i = 0; i < g_dev_free.size(); i++)
            if (g_dev_free[i].device == dev && g_dev_free[i].bytes == cls) {
                const DevBlock b = g_dev_free[i];
                g_dev_cached -= cls;
                g_dev_free.erase(g_dev_free.begin() + (long)i);
                g_dev_live[b.p] = {dev, cls};
                const bool quiet = g_dev_synced[dev] >= b.gen;
                const unsigned long long now = g_dev_gen;
                lk.unlock();
                if (!quiet) {                                       // work queued before the block was parked may still touch it
                    const hipError_t se = hipDeviceSynchronize();
                    if (se != hipSuccess) return se;
                    lk.lock();
                    if (g_dev_synced[dev] < now) g_dev_synced[dev] = now;
                }
                *p = b.p;
                return hipSuccess;
            }
    }
    hipError_t e = malloc_hard(p, cls);
    if (e == hipErrorOutOfMemory && cls > bytes) { (void)hipGetLastError(); e = hipMalloc(p, bytes); if (e == hipSuccess) return e; }   // no room for the rounding: an exact, uncached block
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_dev_mu); g_dev_live[*p] = {dev, cls}; }
    return e;
}
hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    const size_t cap = dev_cache_cap();
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        auto it = g_dev_live.find(p);
        if (it != g_dev_live.end()) {
            const int dev = it->second.first;
            const size_t cls = it->second.second;
            g_dev_live.erase(it);
            if (g_dev_cached + cls <= cap) {
                g_dev_free.push_back({dev, cls, p, ++g_dev_gen});
                g_dev_cached += cls;
                return hipSuccess;
            }
        }
    }
    return hipFree(p);
}
size_t dev_cached_bytes() { std::lock_guard<std::mutex> lk(g_dev_mu); return g_dev_cached; }
bool big_release_all() {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    bool any = false;
    for (;;) {
        DevBlock b;
        {
            std::lock_guard<std::mutex> lk(g_dev_mu);
            if (g_dev_free.empty()) break;
            b = g_dev_free.back();
            g_dev_free.pop_back();
            g_dev_cached -= b.bytes;
        }
        hipSetDevice(b.device);
        hipFree(b.p);
        any = true;
    }
    if (any && have_cur) hipSetDevice(cur);
    return any;
}
}  // namespace lhgt

extern "C" {

int lhgt_digest(lhgt_ctx* ctx, int what, uint64_t mask, uint64_t out[2]) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !out) LHGT_FAIL(LHGT_E_ARG, "null argument");
    const void* p = nullptr;
    uint64_t n = 0, index_base = 0;
    int bytes = 4;
    switch (what) {
        case 0: p = ctx->d_counts; n = ctx->counts_words; break;                                   // packed 2-bit table, as words
        case 1:                                                                                    // per reference position, numbered over ALL indexed
            p = ctx->d_flags; n = ctx->n_pos; bytes = 1;                                           // contigs: the digests of a reference's shards add up
            if (!ctx->all_lens.empty() && !ctx->contigs.empty())                                   // (mod 2^64) to the digest of the whole
                for (uint32_t c = 0; c + 1 < ctx->contigs[0].ref_index && c < ctx->all_lens.size(); c++) index_base += ctx->all_lens[c];
            break;
        case 2: p = ctx->d_peak_kmer; n = ctx->d_peak_kmer ? (1ull << ctx->k) : 0; break;
        case 3: p = ctx->d_loci; n = ctx->n_peaks > 0 ? 2ull * (uint64_t)ctx->id_end : 0; break;
        case 4: p = ctx->d_filter; n = ctx->n_peaks > 0 ? (uint64_t)ctx->id_end : 0; break;         // votes (u32, unclamped)
        default: LHGT_FAIL(LHGT_E_ARG, "digest of what?");
    }
    out[0] = out[1] = 0;
    if (!p || !n) return LHGT_OK;
    if (!ctx->d_digest) LHGT_HIP(lhgt::dev_malloc(&ctx->d_digest, 16));     // one small scratch per context, freed with it
    unsigned long long* d_out = ctx->d_digest;
    LHGT_HIP(hipMemsetAsync(d_out, 0, 16, ctx->stream));
    if (bytes == 1) hipLaunchKernelGGL((digest_kernel<uint8_t>), dim3(8192), dim3(256), 0, ctx->stream, (const uint8_t*)p, n, mask, index_base, d_out);
    else hipLaunchKernelGGL((digest_kernel<uint32_t>), dim3(8192), dim3(256), 0, ctx->stream, (const uint32_t*)p, n, mask, index_base, d_out);
    unsigned long long h[2] = {0, 0};
    hipError_t e1 = hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e1 != hipSuccess || e2 != hipSuccess) LHGT_FAIL(LHGT_E_HIP, "digest copy failed");
    out[0] = h[0];
    out[1] = h[1];
    return LHGT_OK;
}

// Work counters for bench.py's "bytes the implemented algorithm must move" (DESIGN.md 5).  enable = 1: start counting from zero;
// 0: stop; -1: leave as it is.  out (nullable) receives
//   [0] keys phase A's partition brought to its final buckets (valid k-mers x e of the counted mates, minus those applied to the table on
//       the way because a row, a piece or a region was full; the direct kernel reports the upper bound k-mer positions x e)
//   [1] table probes of phase B's probe kernel in the LAST lhgt_ref_scan while counting was on: e per position with a k-mer in the
//       exact form; in the single-first / trio-first forms the hashes ref_flags_lite / ref_flags_trio marked as probed in the
//       per-position state bytes (summed right behind that kernel; the fill of the unsettled tiles comes later and is not in it)
//   [3] probes that went on from the LDS fold to the L2 bitmap (vote_kernel_fold), [4] probes that went on from the bitmap to peak_kmer
//       (vote_kernel_queued / vote_kernel_fold), [5] pairs voted in the lane-per-offset form after the filters (deferred / re-voted)
//   [2] positions phase B's slot-first form followed beyond the list (their slot reads 3: one line of hashes / bases each; [1] then holds
//       their probes), [6], [7] reserved (0).
int lhgt_work_stats(lhgt_ctx* ctx, int enable, unsigned long long out[8]) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || enable < -1 || enable > 1) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if (enable == 1) {
        if (!ctx->d_stats) LHGT_HIP(lhgt::dev_malloc(&ctx->d_stats, 64));
        LHGT_HIP(hipMemsetAsync(ctx->d_stats, 0, 64, ctx->stream));
        memset(ctx->stats_host, 0, sizeof ctx->stats_host);
        ctx->stats_on = true;
        ctx->stats_scan = false;
    } else if (enable == 0) ctx->stats_on = false;
    if (!out) return LHGT_OK;
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!ctx->d_stats) return LHGT_OK;
    // phase B, exact form: e probes per position with a k-mer (the single-first / trio-first forms count theirs on the device, right
    // behind the probe kernel and before the fill of the unsettled tiles: k_scan.hip)
    unsigned long long probes_exact = 0;
    if (ctx->n_peaks >= 0 && ctx->index_resident && ctx->scan_form == 0 && ctx->stats_scan)
        for (const ContigDev& c : ctx->contigs) probes_exact += (unsigned long long)(c.len >= (uint32_t)ctx->k ? c.len - ctx->k + 1 : 0) * ctx->e;
    unsigned long long h[8];
    LHGT_HIP(hipMemcpyAsync(h, ctx->d_stats, 64, hipMemcpyDeviceToHost, ctx->stream));
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; i++) out[i] = h[i] + ctx->stats_host[i];
    if (ctx->scan_form == 0) out[1] = probes_exact;
    return LHGT_OK;
}

int lhgt_device_count(int* n) {
    if (!n) LHGT_FAIL(LHGT_E_ARG, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *n = e == hipSuccess ? c : 0;
    return LHGT_OK;
}

int lhgt_ctx_create(int device, int k, int e, lhgt_ctx** out) {
    if (!out) LHGT_FAIL(LHGT_E_ARG, "null argument");
    *out = nullptr;
    lhgt::entry_context(nullptr);
    if (k < 8 || k > 32) LHGT_FAIL(LHGT_E_ARG, "k = %d outside [8, 32] (hashes are 32-bit, E:1012)", k);
    if (e < 1 || e > 9) LHGT_FAIL(LHGT_E_ARG, "e = %d outside [1, 9]", e);
    if (k * e > LHGT_CODER_SLOTS) LHGT_FAIL(LHGT_E_ARG, "k*e = %d exceeds the %d coder slots (E:1186)", k * e, LHGT_CODER_SLOTS);
    if (device == -1) {  // host-only context: the RNG/coder rows only; every device call fails with E_NO_DEVICE
        lhgt_ctx* c = new lhgt_ctx();
        c->device = -1;
        c->k = k;
        c->e = e;
        memset(c->cc, 0, sizeof c->cc);
        *out = c;
        return LHGT_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "no HIP device visible: this engine has no CPU fallback");
    if (device < 0 || device >= n) LHGT_FAIL(LHGT_E_ARG, "device %d not in [0, %d)", device, n);
    LHGT_HIP(hipSetDevice(device));
    lhgt_ctx* c = new lhgt_ctx();
    c->device = device;
    c->k = k;
    c->e = e;
    memset(c->cc, 0, sizeof c->cc);
    if (const char* sl = getenv("LHGT_SLOT_LIST")) c->sl_mode = atoi(sl) < 0 || atoi(sl) > 2 ? 1 : atoi(sl);
    if (const char* dbg = getenv("LHGT_DEBUG")) c->debug = atoi(dbg);   // lhgt_set_debug's switches for whole-program runs (tests)
    memset(c->rng_state, 0, sizeof c->rng_state);
    c->counts_words = ((size_t)1 << k) / 16;
    // non-blocking: two contexts on one GPU (pipelined samples: one in phase A, the other in B-C) must not meet in the null stream
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreate(&c->ev0);
    if (he == hipSuccess) he = hipEventCreate(&c->ev1);
    if (he == hipSuccess) he = hipEventCreate(&c->ev2);
    if (he == hipSuccess) he = hipEventCreate(&c->ev3);
    if (he == hipSuccess) he = lhgt::dev_malloc(&c->d_counts, c->counts_words * 4);
    if (he == hipSuccess) he = hipMemsetAsync(c->d_counts, 0, c->counts_words * 4, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) {
        set_error("context setup failed: %s", hipGetErrorString(he));
        lhgt_ctx_destroy(c);
        return LHGT_E_HIP;
    }
    *out = c;
    return LHGT_OK;
}

int lhgt_ctx_destroy(lhgt_ctx* c) {
    if (!c) return LHGT_OK;
    lhgt::entry_context(nullptr);       // (an allocation that runs out of memory looks at the calling thread's context: not at this one any more)
    lhgt::sampling_join(c);
    if (c->device < 0) { free(c->rng); delete c; return LHGT_OK; }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    lhgt_pairs_clear(c);
    lhgt::ingest_free(c);
    lhgt_ingest_pool_free(c);
    if (c->h_packed_stage) { (void)hipHostFree(c->h_packed_stage); c->h_packed_stage = nullptr; }
    lhgt::slot_list_drop(c);
    lhgt::vshared_free(c);
    for (void* p : {(void*)c->d_counts, (void*)c->d_index, (void*)c->d_ref_planes, (void*)c->d_contigs, (void*)c->d_tiles, (void*)c->d_flags, (void*)c->d_nzmask, (void*)c->d_tile_good, (void*)c->d_satline, (void*)c->d_active_tiles,
                    (void*)c->d_loci, (void*)c->d_filter, (void*)c->d_tile_count, (void*)c->d_tile_sel, (void*)c->d_rg_buf, (void*)c->d_vote_groups,
                    (void*)c->d_ws_ascii, (void*)c->d_ws_words,
                    (void*)c->d_part_meta, c->d_voted, (void*)c->d_prefilter, (void*)c->d_prefilter_fold, (void*)c->d_revote, (void*)c->d_emit_loci, (void*)c->d_emit_regs, (void*)c->d_digest, (void*)c->d_stats})
        if (p) lhgt::dev_free(p);
    for (void* p : {(void*)c->d_peak_kmer, (void*)c->d_part_keys[0], (void*)c->d_part_keys[1]}) lhgt::dev_free(p);   // large blocks stay with the process (lhgt_common.hpp: dev_free)
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev2) hipEventDestroy(c->ev2);
    if (c->ev3) hipEventDestroy(c->ev3);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    free(c->rng);
    delete c;
    return LHGT_OK;
}

int lhgt_pool_trim(void) {
    lhgt::big_release_all();
    return LHGT_OK;
}

int lhgt_set_thread_emulation(lhgt_ctx* ctx, int threads) {
    if (!ctx || threads < 1 || threads > 99) LHGT_FAIL(LHGT_E_ARG, "thread emulation: 1 (off) .. 99 threads (split_ref holds 100 groups, E:1284)");
    ctx->emu_threads = threads;
    return LHGT_OK;
}

// 0 = the index file's hashes resident (4e bytes per base), 1 = packed bases resident (3/8 byte per base), hashes recomputed by
// phase B.  Takes effect for the next reference made resident; a resident reference of the other form is dropped.
int lhgt_set_reference_form(lhgt_ctx* ctx, int form) {
    if (!ctx || form < 0 || form > 1) LHGT_FAIL(LHGT_E_ARG, "reference form: 0 (index) or 1 (packed)");
    if ((form == 1) == ctx->ref_packed) return LHGT_OK;
    if (ctx->index_resident) {
        LHGT_DEVICE_ENTRY(ctx);
        LHGT_TRY(lhgt::index_layout(ctx, std::vector<uint32_t>()));
        ctx->index_resident = false;
    }
    ctx->ref_packed = form == 1;
    return LHGT_OK;
}

int lhgt_reference_info(lhgt_ctx* ctx, int* form, unsigned long long* resident_bytes) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (form) *form = ctx->ref_packed ? 1 : 0;
    if (resident_bytes) *resident_bytes = !ctx->index_resident ? 0ull : ctx->ref_packed ? 12ull * ctx->ref_plane_words : 4ull * ctx->index_words;
    return LHGT_OK;
}

// profiling / A-B: the context's stream is recreated so that its kernels only run on the CUs whose bits are set (n_words x 32
// bits; n_words = 0: every CU again).  Two contexts with complementary masks share one GPU without sharing a CU (bench.py:
// pipelined_samples).  Nothing may be in flight on the context.
int lhgt_set_cu_mask(lhgt_ctx* ctx, const uint32_t* mask, int n_words) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || n_words < 0 || (n_words && !mask)) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    hipStream_t st = nullptr;
    if (n_words) LHGT_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask));
    else LHGT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    (void)hipStreamDestroy(ctx->stream);
    ctx->stream = st;
    return LHGT_OK;
}

int lhgt_set_debug(lhgt_ctx* ctx, int flags) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    ctx->debug = flags;
    return LHGT_OK;
}

int lhgt_phase_ms(lhgt_ctx* ctx, int phase, float* ms) {
    if (!ctx || !ms || phase < 0 || phase > 3) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    *ms = ctx->phase_ms[phase];
    return LHGT_OK;
}

int lhgt_scan_info(lhgt_ctx* ctx, int* lite, double* frac_slots_at_3, long* n_tiles, long* n_tiles_exact) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (lite) *lite = ctx->scan_slots ? (ctx->scan_form == 2 ? 3 : 4) : ctx->scan_form;   // 0 exact, 1 single-first (lite), 2 trio-first; with the slot list: 3 slot-first (trio-first's), 4 slot-single (single-first's)
    if (frac_slots_at_3) *frac_slots_at_3 = ctx->scan_frac3;
    if (n_tiles) *n_tiles = ctx->n_tiles;
    if (n_tiles_exact) *n_tiles_exact = ctx->scan_lite ? ctx->scan_n_need : ctx->n_tiles;
    return LHGT_OK;
}

// The slot list of the resident reference (k_scan.hip: ref_flags_slots).  mode 0: never build one, and drop the one that exists;
// 1 (default; LHGT_SLOT_LIST): build it before the second sparse-form scan of the same resident reference; 2: before the first;
// -1: leave the mode as it is.  entries / bytes (nullable): the list as it stands (0 = none).
int lhgt_slot_list(lhgt_ctx* ctx, int mode, unsigned long long* entries, unsigned long long* bytes) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || mode < -1 || mode > 2) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if (mode >= 0) ctx->sl_mode = mode;
    if (mode == 0 && ctx->sl_state != 0) {
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        lhgt::slot_list_drop(ctx);
    }
    if (entries) *entries = ctx->sl_state == 1 ? ctx->sl_entries : 0ull;
    if (bytes) *bytes = ctx->sl_state == 1 ? (ctx->d_sl_mid ? 10ull : 6ull) * (ctx->sl_capacity ? ctx->sl_capacity : ctx->sl_entries) + 8ull * (unsigned long long)(ctx->sl_buckets + 1) : 0ull;
    return LHGT_OK;
}

int lhgt_slot_list_build_ms(lhgt_ctx* ctx, double* ms) {
    if (!ctx || !ms) LHGT_FAIL(LHGT_E_ARG, "null argument");
    *ms = ctx->sl_build_ms;
    return LHGT_OK;
}

int lhgt_vote_info(lhgt_ctx* ctx, int* form, int* bitmap_bits, int* three_quarter) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (form) *form = ctx->vote_form;
    int bits = 0;
    for (unsigned long long m = (unsigned long long)ctx->pf_mask + 1ull; m > 1; m >>= 1) bits++;
    if (bitmap_bits) *bitmap_bits = ctx->prefilter_on ? bits : 0;
    if (three_quarter) *three_quarter = ctx->prefilter_on && ctx->pf_q3 ? 1 : 0;
    return LHGT_OK;
}

int lhgt_stream(lhgt_ctx* ctx, void** hip_stream) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !hip_stream) LHGT_FAIL(LHGT_E_ARG, "null argument");
    *hip_stream = (void*)ctx->stream;
    return LHGT_OK;
}

int lhgt_synchronize(lhgt_ctx* ctx) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    return LHGT_OK;
}

}  // extern "C"
