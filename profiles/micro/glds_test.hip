// Semantics probe for __builtin_amdgcn_global_load_lds (gfx950): 16-byte LDS-direct loads with a per-lane
// global source that is only 4-byte aligned; LDS destination = wave-uniform base + lane*16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void k(const float* src, float* out, int shift) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-lane source: row (lane/8) of stride 831 floats, chunk (lane%8) permuted by an XOR -> arbitrary addresses
    const int row = wave * 8 + lane / 8, slot = (lane % 8) ^ (row & 7);
    const float* g = src + (size_t)row * 831 + slot * 4 + shift;
    float* l = lds + wave * 256;                 // wave-uniform LDS base; hardware adds lane*16
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}

int main() {
    const int n = 64 * 831 + 64;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, shift);
        std::vector<float> r(1024);
        hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
            const int row = w * 8 + lane / 8, slot = (lane % 8) ^ (row & 7);
            const float want = (float)(row * 831 + slot * 4 + shift + j);
            if (r[w * 256 + lane * 4 + j] != want) ++bad;
        }
        printf("shift %d: %d mismatches (first vals %g %g %g %g)\n", shift, bad, r[0], r[1], r[2], r[3]);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
