// probe_scalar.hip -- microbenchmark: do random probes through the SCALAR data cache (s_load_dword, wave-uniform addresses)
// add to the ~50 G/s a CU's vector memory path delivers on random 4-byte loads from HBM?  The vector rate is what one TCP's
// ~64 outstanding misses / HBM latency give; the scalar cache is a separate path with its own miss tracking.
// Modes: V = vector gather (64 probes per instruction), S = scalar loads (1 probe per instruction, ILP in flight),
//        M = both in the same wave (every iteration: one gather + ILP scalar loads).
// Build: hipcc -O3 --offload-arch=gfx950 -o probe_scalar probe_scalar.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int MODE, int ILP>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ table, uint32_t mask_words, int iters, uint32_t* sink, uint32_t salt) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(gid >> 6);
    uint32_t acc = 0, sacc = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2) {
            const uint32_t h = mix(gid * 0x9E3779B9u + it * 0x85EBCA6Bu + salt);
            acc += table[h & mask_words];
        }
        if (MODE == 1 || MODE == 2) {
            uint32_t v[ILP];
#pragma unroll
            for (int u = 0; u < ILP; u++) {
                const uint32_t h = mix(wid * 0x9E3779B9u + (it * ILP + u) * 0xC2B2AE35u + salt * 7u);   // wave-uniform (SALU)
                v[u] = table[h & mask_words];                                                            // -> s_load_dword
            }
#pragma unroll
            for (int u = 0; u < ILP; u++) sacc += v[u];
        }
    }
    if (acc + sacc == 0xdeadbeefu) sink[0] = acc;
}

template <int MODE, int ILP>
static void run(const char* name, const uint32_t* table, uint32_t mask_words, uint32_t* sink) {
    const int blocks = 256 * 8, iters = MODE == 1 ? 4096 : 512;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, table, mask_words, 8, sink, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, table, mask_words, iters, sink, 2u);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double waves = blocks * 4.0;
    const double vec = (MODE == 0 || MODE == 2) ? waves * 64 * iters : 0, sca = (MODE == 1 || MODE == 2) ? waves * ILP * iters : 0;
    printf("%-28s ILP %2d  %8.3f ms  vector %7.2f G/s  scalar %7.2f G/s  total %7.2f G probes/s\n", name, ILP, ms, vec / ms / 1e6, sca / ms / 1e6,
           (vec + sca) / ms / 1e6);
}

int main() {
    for (uint64_t bytes : {1ull << 22, 1ull << 28, 1ull << 30, 1ull << 32}) {
        uint32_t* table; uint32_t* sink;
        CK(hipMalloc(&table, bytes)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(table, 1, bytes));
        const uint32_t mask = (uint32_t)(bytes / 4 - 1);
        printf("== table %llu MiB\n", (unsigned long long)(bytes >> 20));
        run<0, 1>("vector gather", table, mask, sink);
        run<1, 4>("scalar loads", table, mask, sink);
        run<1, 8>("scalar loads", table, mask, sink);
        run<1, 16>("scalar loads", table, mask, sink);
        run<2, 4>("gather + scalar", table, mask, sink);
        run<2, 8>("gather + scalar", table, mask, sink);
        run<2, 16>("gather + scalar", table, mask, sink);
        CK(hipFree(table)); CK(hipFree(sink));
    }
    return 0;
}
