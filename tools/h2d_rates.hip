// h2d_rates.hip -- cost of hipHostMalloc and pageable vs pinned H2D rate (ingest staging design).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    size_t n = 1ull << 30;
    void *d, *pin;
    hipMalloc(&d, n);
    double t = now();
    hipHostMalloc(&pin, n, hipHostMallocDefault);
    printf("hipHostMalloc 1 GiB: %.3f s\n", now() - t);
    t = now(); memset(pin, 1, n); printf("first touch of pinned 1 GiB: %.3f s\n", now() - t);
    char* pg = (char*)malloc(n); memset(pg, 1, n);
    for (int r = 0; r < 2; r++) {
        t = now(); hipMemcpy(d, pg, n, hipMemcpyHostToDevice); printf("pageable H2D 1 GiB: %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
        t = now(); hipMemcpy(d, pin, n, hipMemcpyHostToDevice); printf("pinned   H2D 1 GiB: %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    }
    t = now(); memcpy(pin, pg, n); printf("memcpy pageable->pinned 1 GiB (1 thread): %.3f s\n", now() - t);
    t = now(); hipHostRegister(pg, n, hipHostRegisterDefault); printf("hipHostRegister 1 GiB: %.3f s\n", now() - t);
    t = now(); hipMemcpy(d, pg, n, hipMemcpyHostToDevice); printf("registered H2D 1 GiB: %.3f s\n", now() - t);
    return 0;
}
