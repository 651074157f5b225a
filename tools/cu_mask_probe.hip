// cu_mask_probe.hip -- which CUs does bit i of a hipExtStreamCreateWithCUMask mask enable?  Launches a grid of one-wave workgroups
// on masked streams and tallies (XCC_ID, CU id inside the XCD) of the CUs they ran on.
// Build: hipcc -O3 --offload-arch=gfx950 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void where(uint32_t* out) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a little so that the grid spreads over every enabled CU
    uint64_t t0 = clock64();
    while (clock64() - t0 < 200000) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    const int n = 4096;
    uint32_t* d;
    CK(hipMalloc(&d, n * 4));
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, st, d);
    CK(hipStreamSynchronize(st));
    std::vector<uint32_t> h(n);
    CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus;
    int per_xcc[16] = {0};
    for (uint32_t v : h) {
        // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
        const uint32_t key = (v >> 16) << 16 | (v & 0xff00);
        if (cus.insert(key).second) per_xcc[v >> 16]++;
    }
    int bits = 0;
    for (uint32_t w : mask) bits += __builtin_popcount(w);
    printf("%-28s bits=%3d  distinct CUs=%3zu  per XCC:", name, bits, cus.size());
    for (int x = 0; x < 8; x++) printf(" %2d", per_xcc[x]);
    printf("\n");
    CK(hipFree(d));
    CK(hipStreamDestroy(st));
}

int main() {
    std::vector<uint32_t> m(8, 0);
    auto clear = [&] { for (auto& w : m) w = 0; };
    clear(); for (int i = 0; i < 256; i++) m[i >> 5] |= 1u << (i & 31); run("all 256", m);
    clear(); for (int i = 0; i < 32; i++) m[i >> 5] |= 1u << (i & 31); run("bits 0..31", m);
    clear(); for (int i = 0; i < 128; i++) m[i >> 5] |= 1u << (i & 31); run("bits 0..127", m);
    // (masks that leave an XCD without a CU -- bits with i % 8 == 0 only, the even bits -- were tried once: the runtime ignores them
    //  and runs on all 256 CUs; a later process on the same box then died with a GPU hang, so they are not tried again)
    clear(); for (int i = 0; i < 256; i++) if ((i / 8) % 2 == 0) m[i >> 5] |= 1u << (i & 31); run("(i / 8) even", m);
    clear(); for (int i = 0; i < 256; i++) if ((i / 8) < 16) m[i >> 5] |= 1u << (i & 31); run("(i / 8) < 16  (= 0..127)", m);
    clear(); for (int i = 0; i < 256; i++) if ((i % 32) < 16) m[i >> 5] |= 1u << (i & 31); run("(i % 32) < 16", m);
    return 0;
}
