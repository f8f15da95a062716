// Do fp32 MFMA and packed-fp32 VALU FMAs run at the same time on gfx950?  (both peak at 256 flop/clk/CU)
// One wave = NM independent v_mfma_f32_32x32x2_f32 chains and NV v_pk_fma_f32 per MFMA, interleaved.
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_coissue.hip -o mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error line %d\n", __LINE__); return; } } while (0)

template <int NM, int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NM > 0 ? NM : 1];
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)i, a0};
    const float a = a0 + threadIdx.x, b = b0;
    const f32x2 pa = {a * 1e-3f, b * 1e-3f}, pb = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < (NM > 0 ? NM : 1); ++i) {
                if (NM > 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[(i * NV + j) & 15]) : "v"(pa), "v"(pb));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NM, int NV>
void run(int blocks_per_cu) {
    const int cus = 256, iters = 2000;
    float* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    dim3 g(cus * blocks_per_cu), b(256);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NM, NV>), g, b, 0, 0, out, iters, 1.f, 2.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(s));
    hipLaunchKernelGGL((k<NM, NV>), g, b, 0, 0, out, iters, 1.f, 2.f);
    CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
    float ms; CK(hipEventElapsedTime(&ms, s, e));
    const double waves = (double)cus * blocks_per_cu * 4;
    const int nm = NM > 0 ? NM : 1;
    const double mf = waves * iters * 4.0 * NM * 4096.0, vf = waves * iters * 4.0 * nm * NV * 256.0;
    printf("NM=%d NV=%2d waves/SIMD=%d  %8.3f ms  MFMA %6.1f TF  VALU %6.1f TF  total %6.1f TF\n", NM, NV, blocks_per_cu, ms,
           mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
    CK(hipFree(out));
}

int main() {
    for (int w : {1, 2}) {
        run<4, 0>(w);            // MFMA alone
        run<0, 16>(w);           // VALU alone
        run<4, 2>(w); run<4, 4>(w); run<4, 8>(w); run<4, 12>(w); run<4, 16>(w);
    }
    return 0;
}
