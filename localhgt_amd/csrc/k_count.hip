// k_count.hip -- phase A: saturating k-mer count of the resident reads (read_fastq, E:981-1107).
//
// Table: 2^k slots of 2 bits, 16 per u32 word; slot h holds min(3, occurrences of h), which is
// what the reference's `if (T[h] < 3) T[h]++` (E:1082-1084) yields without its data race and is
// independent of the order in which reads arrive.
#include "lhgt_hash.hpp"

namespace lhgt {

__device__ __forceinline__ void sat_inc(uint32_t* __restrict__ T, uint32_t h) {
    uint32_t* w = T + (h >> 4);
    uint32_t sh = (h & 15u) * 2u;
    // a stale (smaller) value only costs one extra CAS round: fields never decrease
    uint32_t old = __builtin_nontemporal_load(w);
    while (((old >> sh) & 3u) != 3u) {
        uint32_t seen = atomicCAS(w, old, old + (1u << sh));
        if (seen == old) break;
        old = seen;
    }
}

// One wave per read; lane l takes k-mer offsets l, l+64, ...  All e probes of a lane's k-mer
// are independent, so a wave keeps up to 64*e table operations in flight.
// n_as_base: count_diff_kmer.cpp's `bool` coder (C:155-160) -- a non-ACGT base is not rejected: it codes 1 in every projection
// on the forward strand (= A: both code bits 0, as pack_bases stores it) and, its complement being the NUL byte, 1 in every
// projection on the reverse strand too (= the complement of T: both bits 1)
__global__ void __launch_bounds__(256) count_direct(ReadBatchDev b, HashParams hp, uint32_t* __restrict__ counts, int n_as_base) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long n_waves = ((long)gridDim.x * blockDim.x) >> 6;
    const long n_reads = 2 * b.n_pairs;
    for (long r = wave; r < n_reads; r += n_waves) {
        const int m = (int)(r & 1);
        const long p = r >> 1;
        if (b.flags && !((b.flags[p] >> m) & 1)) continue;  // quirk Q4, thread-chunk emulation
        const int len = b.len[m][p];
        const int nk = len - hp.k + 1;
        if (nk <= 0) continue;
        const int wpr = ((len + 31) >> 5) + 1;
        const uint32_t* rec = b.words + b.off[m][p];
        for (int j = lane; j < nk; j += 64) {
            const uint32_t wnb = plane_window(rec + 2 * wpr, j, hp.k);
            if (wnb != 0 && !n_as_base) continue;  // a non-ACGT base voids the k-mer (E:1065-1069)
            uint32_t whi = plane_window(rec, j, hp.k), wlo = plane_window(rec + wpr, j, hp.k);
            uint32_t rhi = brev_k(whi | wnb, hp.k), rlo = brev_k(wlo | wnb, hp.k);   // wnb != 0 only under n_as_base
            for (int i = 0; i < hp.e; i++) sat_inc(counts, hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]));
        }
    }
}

// per-slot saturating add of another table slice: min(3, a+b) on 2-bit fields, 16 per word
__global__ void __launch_bounds__(256) counts_merge_kernel(uint32_t* __restrict__ mine, const uint32_t* __restrict__ other, size_t n_words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_words; i += stride) {
        uint32_t a = mine[i], c = other[i];
        // widen to 4-bit lanes: even and odd fields separately, add, clamp each nibble to 3
        uint32_t ae = a & 0x33333333u, ao = (a >> 2) & 0x33333333u;
        uint32_t ce = c & 0x33333333u, co = (c >> 2) & 0x33333333u;
        uint32_t se = ae + ce, so = ao + co;  // each nibble 0..6
        uint32_t oe = ((se >> 2) & 0x11111111u) * 3u, oo = ((so >> 2) & 0x11111111u) * 3u;  // nibble >= 4 -> 3
        se = (se | oe) & 0x33333333u;
        so = (so | oo) & 0x33333333u;
        mine[i] = se | (so << 2);
    }
}

__global__ void __launch_bounds__(256) counts_expand_u8(const uint32_t* __restrict__ T, uint64_t first, uint64_t n, uint8_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = first + i;
    out[i] = (T[h >> 4] >> ((h & 15u) * 2u)) & 3u;
}

__global__ void __launch_bounds__(256) counts_hist_kernel(const uint32_t* __restrict__ T, size_t n_words, uint64_t slots_in_last,
                                                          unsigned long long* __restrict__ hist) {
    __shared__ unsigned long long sh[4];
    if (threadIdx.x < 4) sh[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long c[4] = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_words; i += stride) {
        uint32_t w = T[i];
        int fields = (i == n_words - 1) ? (int)slots_in_last : 16;
        for (int f = 0; f < fields; f++) c[(w >> (2 * f)) & 3u]++;
    }
    for (int v = 0; v < 4; v++) atomicAdd(&sh[v], c[v]);
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

int lhgt_count_kmers(lhgt_ctx* ctx) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder: load or build the index first");
    LHGT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    // Which kernel: the radix partition needs many table slices to spread over the chip (16384 at k = 32) and wins from
    // k = 26 up (bench workload: 102 vs 117 ms at k = 26, 60 vs 301 ms at k = 32); for smaller k the table is cache
    // resident and saturates at once, so the direct kernel's pre-check load makes it read-mostly (45 vs 1021 ms at k = 21).
    const int mode = ctx->count_compat ? 0 : ctx->count_mode >= 0 ? ctx->count_mode : (ctx->k >= 26 ? 1 : 0);   // the compat coder lives in the direct kernel only
    for (ReadBatch& b : ctx->batches) {
        ctx->counts_touched = true;
        if (b.counted) { b.counted = false; continue; }   // the loader counted it already (lhgt_set_count_on_load): skipped this once
        if (mode == 1) {
            LHGT_TRY(lhgt_count_batch_partitioned(ctx, b));
            continue;
        }
        long waves = 2 * b.d.n_pairs;
        long blocks = (waves + 3) / 4;
        if (blocks > 256L * 8 * 4) blocks = 256L * 8 * 4;
        hipLaunchKernelGGL(count_direct, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, b.d, ctx->hp, ctx->d_counts, ctx->count_compat ? 1 : 0);
        if (ctx->stats_on) ctx->stats_host[0] += b.n_kmers * (unsigned long long)ctx->e;   // lhgt_work_stats: upper bound (k-mers with an N are skipped)
    }
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    LHGT_HIP(hipEventSynchronize(ctx->ev1));
    LHGT_HIP(hipEventElapsedTime(&ctx->phase_ms[0], ctx->ev0, ctx->ev1));
    ctx->phase_ms[0] += ctx->count_on_load_ms;
    ctx->count_on_load_ms = 0.f;
    return LHGT_OK;
}

}  // extern "C"

// phase A of ONE resident batch, asynchronously on the context's stream (the loader's count-on-load); t0 / t1 bracket it
int lhgt_count_one_batch_async(lhgt_ctx* ctx, lhgt::ReadBatch& b, hipEvent_t t0, hipEvent_t t1) {
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder: load or build the index before loading reads with count-on-load");
    const int mode = ctx->count_compat ? 0 : ctx->count_mode >= 0 ? ctx->count_mode : (ctx->k >= 26 ? 1 : 0);
    LHGT_HIP(hipEventRecord(t0, ctx->stream));
    if (mode == 1) LHGT_TRY(lhgt_count_batch_partitioned(ctx, b));
    else {
        long waves = 2 * b.d.n_pairs;
        long blocks = (waves + 3) / 4;
        if (blocks > 256L * 8 * 4) blocks = 256L * 8 * 4;
        hipLaunchKernelGGL(count_direct, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, b.d, ctx->hp, ctx->d_counts, ctx->count_compat ? 1 : 0);
        LHGT_HIP(hipGetLastError());
    }
    LHGT_HIP(hipEventRecord(t1, ctx->stream));
    b.counted = true;
    ctx->counts_touched = true;
    return LHGT_OK;
}

extern "C" {

int lhgt_set_count_on_load(lhgt_ctx* ctx, int on) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    ctx->count_on_load = on != 0;
    return LHGT_OK;
}

int lhgt_set_count_mode(lhgt_ctx* ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) LHGT_FAIL(LHGT_E_ARG, "count mode must be -1 (by k), 0 (direct) or 1 (partitioned)");
    ctx->count_mode = mode;
    return LHGT_OK;
}

int lhgt_set_count_compat(lhgt_ctx* ctx, int on) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    ctx->count_compat = on != 0;
    return LHGT_OK;
}

int lhgt_counts_clear(lhgt_ctx* ctx) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    LHGT_HIP(hipMemsetAsync(ctx->d_counts, 0, ctx->counts_words * 4, ctx->stream));
    for (ReadBatch& b : ctx->batches) b.counted = false;
    ctx->count_on_load_ms = 0.f;
    ctx->counts_touched = false;
    return LHGT_OK;
}

int lhgt_counts_buffer(lhgt_ctx* ctx, void** dev_ptr, size_t* bytes) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !dev_ptr || !bytes) LHGT_FAIL(LHGT_E_ARG, "null argument");
    ctx->counts_touched = true;      // the caller may write through the pointer (the count-table exchange does)
    *dev_ptr = ctx->d_counts;
    *bytes = ctx->counts_words * 4;
    return LHGT_OK;
}

int lhgt_counts_merge(lhgt_ctx* ctx, const void* dev_other, size_t byte_offset, size_t bytes) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !dev_other) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (byte_offset % 4 || bytes % 4 || byte_offset + bytes > ctx->counts_words * 4)
        LHGT_FAIL(LHGT_E_ARG, "merge range [%zu,+%zu) outside the table or not word aligned", byte_offset, bytes);
    size_t n = bytes / 4;
    if (!n) return LHGT_OK;
    ctx->counts_touched = true;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(counts_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->d_counts + byte_offset / 4,
                       (const uint32_t*)dev_other, n);
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    return LHGT_OK;
}

int lhgt_counts_export_u8(lhgt_ctx* ctx, uint64_t first_slot, uint64_t n_slots, uint8_t* out) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !out) LHGT_FAIL(LHGT_E_ARG, "null argument");
    uint64_t total = 1ull << ctx->k;
    if (first_slot + n_slots > total) LHGT_FAIL(LHGT_E_ARG, "slot range outside the table");
    const uint64_t CH = 1ull << 28;
    uint8_t* d_tmp;
    LHGT_HIP(lhgt::dev_malloc(&d_tmp, n_slots < CH ? n_slots : CH));
    for (uint64_t o = 0; o < n_slots; o += CH) {
        uint64_t n = n_slots - o < CH ? n_slots - o : CH;
        hipLaunchKernelGGL(counts_expand_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_counts,
                           first_slot + o, n, d_tmp);
        hipError_t e1 = hipMemcpyAsync(out + o, d_tmp, n, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        if (e1 != hipSuccess || e2 != hipSuccess) { lhgt::dev_free(d_tmp); LHGT_FAIL(LHGT_E_HIP, "export copy failed"); }
    }
    lhgt::dev_free(d_tmp);
    return LHGT_OK;
}

int lhgt_counts_histogram(lhgt_ctx* ctx, uint64_t out[4]) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !out) LHGT_FAIL(LHGT_E_ARG, "null argument");
    unsigned long long* d_h;
    LHGT_HIP(lhgt::dev_malloc(&d_h, 32));
    LHGT_HIP(hipMemsetAsync(d_h, 0, 32, ctx->stream));
    uint64_t slots = 1ull << ctx->k;
    uint64_t in_last = slots % 16 ? slots % 16 : 16;
    hipLaunchKernelGGL(counts_hist_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->d_counts, ctx->counts_words, in_last, d_h);
    unsigned long long h[4];
    LHGT_HIP(hipMemcpyAsync(h, d_h, 32, hipMemcpyDeviceToHost, ctx->stream));
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    lhgt::dev_free(d_h);
    for (int i = 0; i < 4; i++) out[i] = h[i];
    return LHGT_OK;
}

}  // extern "C"
