// Ceiling probe: v_mfma_f32_32x32x2_f32 issue rate with W waves per SIMD, NACC accumulators, and
// optional ds_read_b128 operand traffic.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        float4 fa = make_float4(a, a, a, a), fb = make_float4(b, b, b, b);
        if (LDS) {
            fa = *reinterpret_cast<const float4*>(lds + ((lane * 36 + (it & 7) * 4) & 4095 & ~3));
            fb = *reinterpret_cast<const float4*>(lds + ((lane * 36 + 2048 + (it & 7) * 4) & 4095 & ~3));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(((float*)&fa)[t], ((float*)&fb)[t], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(int blocks_per_cu, const char* tag) {
    int cus = 256, iters = 2000;
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    dim3 g(cus * blocks_per_cu), b(256);
    hipLaunchKernelGGL((k<NACC, LDS>), g, b, 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL((k<NACC, LDS>), g, b, 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)cus * blocks_per_cu * 4 /*waves*/ * iters * 4.0 * NACC * 4096.0;
    printf("%-28s blocks/CU=%d  %.3f ms  %.1f TF/s\n", tag, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) run<1, false>(w, "NACC=1 regs");
    for (int w = 1; w <= 4; ++w) run<2, false>(w, "NACC=2 regs");
    for (int w = 1; w <= 3; ++w) run<4, false>(w, "NACC=4 regs");
    for (int w = 1; w <= 4; ++w) run<2, true>(w, "NACC=2 +2 ds_read_b128/8mfma");
    for (int w = 1; w <= 3; ++w) run<4, true>(w, "NACC=4 +2 ds_read_b128/16mfma");
    return 0;
}
