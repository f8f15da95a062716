// Micro-benchmark (not shipped): ablations of the planes GEMM's ring kernel (csrc/gemm_p3.hip, P3_ABL bits) on the layer-0
// forward shape, and the in-kernel clock (s_memtime / s_memrealtime around the K loop).
// build: for a in 0 1 2 4 8 16 ...; do hipcc -O3 --offload-arch=gfx950 -std=c++17 -DP3_ABL=$a profiles/micro/gemm_p3_abl.hip \
//            gnn-tableextraction_amd/csrc/gte_core.hip -o profiles/micro/abl/p3_$a; done
// run:   GTE_P3_NT_CFG=c profiles/micro/abl/p3_a [M N K]
#include "../../gnn-tableextraction_amd/csrc/gemm_p3.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void fill_kernel(float* x, long long n, unsigned seed) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000);        // uniform [-1, 1)
}

int main(int argc, char** argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 24437, N = argc > 2 ? atoll(argv[2]) : 512, K = argc > 3 ? atoll(argv[3]) : 831;
    float *a, *b, *c;
    char *pa, *pb;
    long long* stamps;
    const long long lda = gte_p3_row_bytes(K);
    hipMalloc(&a, M * K * 4); hipMalloc(&b, N * K * 4); hipMalloc(&c, M * N * 4);
    hipMalloc(&pa, M * lda); hipMalloc(&pb, N * lda); hipMalloc(&stamps, 4096 * 16);
    hipMemset(stamps, 0, 4096 * 16);
    fill_kernel<<<(unsigned)((M * K + 255) / 256), 256>>>(a, M * K, 1u);
    fill_kernel<<<(unsigned)((N * K + 255) / 256), 256>>>(b, N * K, 2u);
    gte_p3_from_f32(a, K, M, K, 0, pa, lda, nullptr);
    gte_p3_from_f32(b, K, N, K, 0, pb, lda, nullptr);
    P3Gemm p = {};
    p.A1 = pa; p.lda1 = lda; p.KB1 = (int)((K + 15) / 16); p.B = pb; p.ldb = lda; p.C = c; p.ldc = N; p.M = (int)M; p.N = (int)N;
    p.splits = 1; p.slab = reinterpret_cast<float*>(stamps);
    p.bsa1 = p.bsb = 96;
    if (getenv("P3_BLOCKED")) {          // block-major addressing (timing only: the images are still row-major, results are garbage)
        p.lda1 = 96; p.bsa1 = M * 96; p.ldb = 96; p.bsb = N * 96;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) launch_nt(p, nullptr);
    hipDeviceSynchronize();
    // sustained: ~0.5 s of back-to-back launches, then the timed ones
    for (int i = 0; i < 3000; ++i) launch_nt(p, nullptr);
    hipEventRecord(e0);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) launch_nt(p, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, gf = 2.0 * M * N * K * 1e-9;
    printf("%sabl %d cfg %s %lldx%lldx%lld: %.1f us  %.1f TF fp32-eq (%.0f TF bf16)", getenv("P3_BLOCKED") ? "blocked " : "", P3_ABL, getenv("GTE_P3_NT_CFG") ? getenv("GTE_P3_NT_CFG") : "auto",
           M, N, K, us, gf / us * 1e3, 6 * gf / us * 1e3);
#if P3_ABL & 16
    std::vector<long long> h(4096 * 2);
    hipMemcpy(h.data(), stamps, 4096 * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < 4096; ++i)
        if (h[2 * i + 1] > 0) { ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i]); }
    if (!ghz.empty()) {
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        printf("  in-kernel clock median %.3f GHz (min %.3f max %.3f), K-loop cycles median %.0f over %zu workgroups", ghz[ghz.size() / 2],
               ghz.front(), ghz.back(), cyc[cyc.size() / 2], ghz.size());
    }
#endif
    printf("\n");
    return 0;
}
