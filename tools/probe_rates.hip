// probe_rates.hip -- microbenchmark: random 4-byte probes into an HBM table on MI355X.
// Decides the count-table layout (plain gather vs returning atomicOr vs load+CAS) and shows
// how the rate moves with table size (L2 4 MiB/XCD, Infinity Cache 256 MiB, HBM).
// Build: hipcc -O3 --offload-arch=gfx950 -o probe_rates probe_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
enum { LOAD = 0, OR_NORET = 1, OR_RET = 2, LOAD_CAS = 3, LOAD_U8 = 4, ADD_NORET = 5, STORE = 6, MAX_NORET = 7 };

template <int MODE, int ILP>
__global__ void __launch_bounds__(256) probe(uint32_t* __restrict__ table, uint64_t mask_words, int iters, uint32_t* sink, uint32_t salt) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t h[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) h[u] = mix(gid * 0x9E3779B9u + (it * ILP + u) * 0x85EBCA6Bu + salt);
        if (MODE == LOAD) {
#pragma unroll
            for (int u = 0; u < ILP; u++) acc += table[h[u] & mask_words];
        } else if (MODE == LOAD_U8) {
#pragma unroll
            for (int u = 0; u < ILP; u++) acc += ((const uint8_t*)table)[(uint64_t)h[u] & (mask_words * 4 + 3)];
        } else if (MODE == OR_NORET) {
#pragma unroll
            for (int u = 0; u < ILP; u++) atomicOr(&table[h[u] & mask_words], 1u << (h[u] >> 27));
        } else if (MODE == ADD_NORET) {
#pragma unroll
            for (int u = 0; u < ILP; u++) atomicAdd(&table[h[u] & mask_words], 1u);
        } else if (MODE == STORE) {
#pragma unroll
            for (int u = 0; u < ILP; u++) table[h[u] & mask_words] = h[u];
        } else if (MODE == MAX_NORET) {
#pragma unroll
            for (int u = 0; u < ILP; u++) atomicMax(&table[h[u] & mask_words], h[u]);
        } else if (MODE == OR_RET) {
#pragma unroll
            for (int u = 0; u < ILP; u++) acc += atomicOr(&table[h[u] & mask_words], 1u << (h[u] >> 27));
        } else if (MODE == LOAD_CAS) {
            uint32_t w[ILP];
#pragma unroll
            for (int u = 0; u < ILP; u++) w[u] = table[h[u] & mask_words];
#pragma unroll
            for (int u = 0; u < ILP; u++) {
                uint32_t sh = (h[u] >> 28) * 2, old = w[u];
                while (((old >> sh) & 3u) != 3u) {
                    uint32_t seen = atomicCAS(&table[h[u] & mask_words], old, old + (1u << sh));
                    if (seen == old) break;
                    old = seen;
                }
                acc += old;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int ILP>
void run(const char* name, uint32_t* table, uint64_t bytes, uint32_t* sink) {
    const int blocks = 256 * 8, threads = 256, iters = 64;
    uint64_t mask_words = bytes / 4 - 1;
    double probes = (double)blocks * threads * iters * ILP;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemsetAsync(table, 0, bytes, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((probe<MODE, ILP>), dim3(blocks), dim3(threads), 0, 0, table, mask_words, iters, sink, 17u * rep);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-10s ILP=%d table=%6.0f MiB  %8.3f ms  %7.2f Gprobe/s  (x64B = %6.2f TB/s)\n", name, ILP, bytes / 1048576.0, best,
           probes / best / 1e6, probes * 64 / best / 1e9);
    fflush(stdout);
}

int main() {
    uint64_t max_bytes = 16ull << 30;
    uint32_t *table, *sink;
    CK(hipMalloc(&table, max_bytes)); CK(hipMalloc(&sink, 4));
    if (getenv("STORES_ONLY")) {
        for (uint64_t s : {1ull << 30, 16ull << 30}) {
            run<STORE, 12>("store", table, s, sink);
            run<MAX_NORET, 12>("max_noret", table, s, sink);
            run<LOAD, 12>("load", table, s, sink);
        }
        return 0;
    }
    uint64_t sizes[] = {4ull << 20, 32ull << 20, 128ull << 20, 512ull << 20, 1ull << 30, 4ull << 30, 16ull << 30};
    for (uint64_t s : sizes) {
        run<LOAD, 4>("load", table, s, sink);
        run<LOAD, 12>("load", table, s, sink);
        run<LOAD_U8, 12>("load_u8", table, s, sink);
        run<OR_NORET, 12>("or_noret", table, s, sink);
        run<ADD_NORET, 12>("add_noret", table, s, sink);
        run<OR_RET, 12>("or_ret", table, s, sink);
        run<LOAD_CAS, 12>("load_cas", table, s, sink);
    }
    return 0;
}
