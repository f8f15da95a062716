// Which copy shape reaches the best HBM rate on this box?  (yardstick for the aggregation kernel's load/store pattern)
//   variants: grid-stride vs contiguous chunk per workgroup; 1 / 4 / 8 16-byte accesses in flight per lane; plain vs
//   non-temporal loads / stores.  build: hipcc -O3 --offload-arch=gfx950 copy_variants.hip -o copy_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %d\n", __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NTL, bool NTS, bool CHUNK>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t T = (size_t)gridDim.x * 256;
    size_t i, step;
    if (CHUNK) {           // workgroup w owns [w * per, (w + 1) * per); lanes stride by 256 inside
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        const size_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
        for (i = lo + threadIdx.x; i + (U - 1) * 256 < hi; i += U * 256) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * 256) : a[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], b + i + u * 256); else b[i + u * 256] = v[u]; }
        }
        for (; i < hi; i += 256) b[i] = a[i];
        return;
    }
    for (i = blockIdx.x * (size_t)256 + threadIdx.x; i + (U - 1) * T < n; i += U * T) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * T) : a[i + u * T];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], b + i + u * T); else b[i + u * T] = v[u]; }
    }
    for (; i < n; i += T) b[i] = a[i];
}
template <int U, bool NTL, bool NTS, bool CHUNK>
int run(const char* name, const f4* a, f4* b, size_t n, hipEvent_t e0, hipEvent_t e1) {
    for (int grid : {1024, 2048, 4096, 16384}) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_copy<U, NTL, NTS, CHUNK>), dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_copy<U, NTL, NTS, CHUNK>), dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-34s grid %6d: %7.1f us %5.2f TB/s\n", name, grid, ms * 1e3, 2.0 * n * 16 / ms / 1e9);
    }
    return 0;
}
int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? atol(argv[1]) : 2048;
    const size_t bytes = mib << 20, n = bytes / 16;
    f4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<1, false, false, false>("stride U1", a, b, n, e0, e1);
    run<4, false, false, false>("stride U4", a, b, n, e0, e1);
    run<8, false, false, false>("stride U8", a, b, n, e0, e1);
    run<4, false, true, false>("stride U4 nt-store", a, b, n, e0, e1);
    run<4, true, true, false>("stride U4 nt-load nt-store", a, b, n, e0, e1);
    run<4, true, false, false>("stride U4 nt-load", a, b, n, e0, e1);
    run<4, false, false, true>("chunk U4", a, b, n, e0, e1);
    run<8, false, false, true>("chunk U8", a, b, n, e0, e1);
    run<4, false, true, true>("chunk U4 nt-store", a, b, n, e0, e1);
    run<4, true, true, true>("chunk U4 nt-load nt-store", a, b, n, e0, e1);
    run<8, true, true, true>("chunk U8 nt-load nt-store", a, b, n, e0, e1);
    return 0;
}
