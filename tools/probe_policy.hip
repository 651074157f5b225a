// probe_policy.hip -- do random 4-byte probes cost a 128-B line of HBM bandwidth, and does a cache policy change it?
// Random loads from a 16 GiB table with each gfx950 load modifier combination; rate in G probes/s.
// Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum to see the request sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

#define LOADER(NAME, MODS)                                                                          \
    __device__ __forceinline__ uint32_t NAME(const uint32_t* p) {                                    \
        uint32_t v;                                                                                 \
        asm volatile("global_load_dword %0, %1, off " MODS : "=v"(v) : "v"(p) : "memory");          \
        return v;                                                                                   \
    }
LOADER(ld_plain, "")
LOADER(ld_nt, "nt")
LOADER(ld_sc0, "sc0")
LOADER(ld_sc1, "sc1")
LOADER(ld_sc0sc1, "sc0 sc1")
LOADER(ld_sc1nt, "sc1 nt")
LOADER(ld_all, "sc0 sc1 nt")
LOADER(ld_sc0nt, "sc0 nt")

template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ table, uint64_t mask_words, int iters, uint32_t* sink, uint32_t salt) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    constexpr int ILP = 8;
    for (int it = 0; it < iters; it++) {
        uint32_t v[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) {
            const uint32_t* p = table + (mix(gid * 0x9E3779B9u + (it * ILP + u) * 0x85EBCA6Bu + salt) & mask_words);
            if (MODE == 0) v[u] = ld_plain(p);
            if (MODE == 1) v[u] = ld_nt(p);
            if (MODE == 2) v[u] = ld_sc0(p);
            if (MODE == 3) v[u] = ld_sc1(p);
            if (MODE == 4) v[u] = ld_sc0sc1(p);
            if (MODE == 5) v[u] = ld_sc1nt(p);
            if (MODE == 6) v[u] = ld_all(p);
            if (MODE == 7) v[u] = ld_sc0nt(p);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < ILP; u++) acc += v[u];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
void run(const char* name, uint32_t* table, uint64_t bytes, uint32_t* sink) {
    const int blocks = 256 * 8, threads = 256, iters = 96;
    double probes = (double)blocks * threads * iters * 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, table, bytes / 4 - 1, iters, sink, 17u * rep);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-12s table=%6.0f MiB  %8.3f ms  %7.2f Gprobe/s\n", name, bytes / 1048576.0, best, probes / best / 1e6);
    fflush(stdout);
}

int main() {
    uint64_t bytes = 16ull << 30;
    uint32_t *table, *sink;
    CK(hipMalloc(&table, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(table, 1, bytes));
    // 4 MiB and below: the L2-resident regime of the vote kernel's Bloom bitmap (round 3: its probes scale with the number of CUs,
    // tools/cu_share.py -- is the limit the 128-B line fill into each CU's L1, and does a load that bypasses the L1 lift it?)
    for (uint64_t s : {16ull << 30, 1ull << 30, 64ull << 20, 16ull << 20, 4ull << 20, 1ull << 20, 256ull << 10}) {
        run<0>("plain", table, s, sink); run<1>("nt", table, s, sink); run<2>("sc0", table, s, sink); run<3>("sc1", table, s, sink);
        run<4>("sc0 sc1", table, s, sink); run<5>("sc1 nt", table, s, sink); run<6>("sc0 sc1 nt", table, s, sink); run<7>("sc0 nt", table, s, sink);
    }
    return 0;
}
