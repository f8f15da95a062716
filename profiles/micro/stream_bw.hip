// Achievable HBM bandwidth yardsticks on this box: float4 copy, read-only, write-only (grid-stride, 256-thread blocks).
// build: hipcc -O3 --offload-arch=gfx950 stream_bw.hip -o stream_bw ; run: ./stream_bw [MiB per array]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %d\n", __LINE__); return 1; } } while (0)
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read(const float4* __restrict__ a, float* __restrict__ sink, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) sink[0] = s;
}
__global__ void k_write(float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? atol(argv[1]) : 2048;
    const size_t bytes = mib << 20, n = bytes / 16;
    float4 *a, *b; float* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 8192, 32768}) {
        for (int what = 0; what < 3; ++what) {
            auto launch = [&]() {
                if (what == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                if (what == 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n);
                if (what == 2) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n);
            };
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e0));
            const int reps = 20;
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            const double moved = (what == 0 ? 2.0 : 1.0) * bytes;
            printf("%-5s %5zu MiB/array grid %6d: %8.1f us  %6.2f TB/s\n", what == 0 ? "copy" : what == 1 ? "read" : "write", mib, grid, ms * 1e3, moved / ms / 1e9);
        }
    }
    return 0;
}
