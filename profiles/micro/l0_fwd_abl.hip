// Micro-benchmark (not shipped): the layer-0 forward of the train step as the step runs it -- [x | cached ahn] W^T through the
// batch's row map, K = 2 x 831, N = 256, LayerNorm + ReLU epilogue (gte_gemm_p3_nt_rows2_ln_fwd) -- with the P3_ABL ablations of
// the loader-wave kernel (csrc/gemm_p3.hip), and a REQUESTS-ONLY kernel that measures what a CU takes in through LDS-DMA for a
// given fetch shape (bytes per row and request, row-major / block-major weights, cache policy, requests in flight).
// build: profiles/debug/l0_abl_build.sh      run: profiles/debug/l0_abl_run.sh   (-> profiles/r06/l0_fwd_ablation.txt)
#include "../../gnn-tableextraction_amd/csrc/gemm_p3.hip"

#ifndef ABL_NL
#define ABL_NL 4
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void fill_kernel(float* x, long long n, unsigned seed) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000);        // uniform [-1, 1)
}

// ---- requests only ------------------------------------------------------------------------------------------------------
// One workgroup of 4 loader waves per CU, no consumer.  A macro step = 64 k (four P3 blocks, 384 bytes per row): the A tile's
// rowsA x 384 bytes in requests of Q x 96 bytes per row, the weights' 256 x 384 bytes as four 16-k stages (row-major: 96-byte runs
// at row stride ldb; block-major: one contiguous 24 KB run per stage) or in 384-byte runs.  At most W requests of a wave in flight.
struct DmaArgs {
    const char* A; const char* A2; const char* B; const int* rowmap;
    int M, rowsA, KB;                 // KB: 16-k blocks per A image row (52)
    long long lda, res_bytes;
    int rowsB; long long ldb, bsb;    // row-major: ldb = row stride, bsb = 96; block-major: ldb = 96, bsb = rowsB * 96
    int skipA, skipB;
    long long* stamps;
};

template <int AUX>
__device__ __forceinline__ void dma16x(__amdgpu_buffer_rsrc_t srd, char* dst, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)dst, 16, voffset, 0, 0, AUX);
}
template <int W>
__device__ __forceinline__ void window_wait() {
#define GTE_WW(n) else if constexpr (W - 1 == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    if constexpr (W - 1 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GTE_WW(5); GTE_WW(8); GTE_WW(11); GTE_WW(17); GTE_WW(23); GTE_WW(26); GTE_WW(35); GTE_WW(47); GTE_WW(53);
#undef GTE_WW
}

// Q: 1 / 2 / 4 = 96 / 192 / 384 bytes per A row and request group; BM: 0 row-major 96-byte runs, 1 block-major, 2 row-major 384-byte runs
// Q = 8: the A tile in 96-byte pieces as Q = 1, but the FOUR blocks of a row group are requested back to back by the same wave (what a
// block-major placement of a 64-deep A slot in LDS would issue): do the requests of one line coalesce in the vector L1?
template <int Q, int BMODE, int W, int NTA, int NTB>
__global__ void __launch_bounds__(256) dma_intake_kernel(const DmaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];          // 128 KB, used as a ring of 1 KB pieces
    const int lane = threadIdx.x & 63;
    const int lw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lb = gte_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (int)lb * p.rowsA;
    const int rowsA = min(p.rowsA, p.M - m0);
    constexpr int QQ = Q == 8 ? 1 : Q;
    constexpr int NL = 4, CPR = 6 * QQ;                                 // 16-byte chunks per A row and request group
    const int a_inst = p.rowsA * 96 / 1024;                             // A requests per 16-k stage (the whole workgroup): 9 at 96 rows
    // this lane's resident row for A request (rpart, a): up to Q x 3 per wave
    constexpr int NA = 3;
    int rowoff[QQ][NA];
    static_for<QQ>([&](auto RP) {
        static_for<NA>([&](auto AI) {
            constexpr int rp = decltype(RP)::value, ai = decltype(AI)::value;
            const int a = ai * NL + lw;
            const int s = (rp * a_inst + a) * 64 + lane, row = s / CPR, chunk = s - row * CPR;
            const int rr = (a < a_inst && row < rowsA) ? p.rowmap[m0 + row] : -1;
            rowoff[rp][ai] = rr < 0 ? (int)0xfffffff0u : (int)((unsigned)rr * (unsigned)p.lda + (unsigned)(chunk * 16));
        });
    });
    const int macros = 2 * p.KB / 4;                                    // 26
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned cnt = 0;
    for (int mac = 0; mac < macros; ++mac) {
        const bool seg = mac >= macros / 2;
        const int kb0 = (seg ? mac - macros / 2 : mac) * 4;            // first 16-k block of the macro step inside its image
        const __amdgpu_buffer_rsrc_t sa =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(seg ? p.A2 : p.A), 0, p.skipA ? 0 : (int)(unsigned)p.res_bytes, SRD_FLAGS);
        static_for<4>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int kp = j / QQ, rp = j % QQ;
            // A: request group (kp, rp)
            if constexpr (Q == 8) {                                      // row group j of this wave: its four blocks back to back
                if constexpr (j < NA) {
                    if (j * NL + lw < a_inst) {
                        static_for<4>([&](auto KB_) {
                            dma16x<NTA ? 2 : 0>(sa, lds + ((cnt * NL + lw) & 127) * 1024, rowoff[0][j] + (kb0 + decltype(KB_)::value) * 96);
                            ++cnt;
                            window_wait<W>();
                        });
                    }
                }
            } else {
                static_for<NA>([&](auto AI) {
                    constexpr int ai = decltype(AI)::value;
                    if (ai * NL + lw < a_inst) {
                        dma16x<NTA ? 2 : 0>(sa, lds + ((cnt * NL + lw) & 127) * 1024, rowoff[rp][ai] + (kb0 + kp * QQ) * 96);
                        ++cnt;
                        window_wait<W>();
                    }
                });
            }
            // B: stage j of the macro step (256 rows x 96 bytes = 24 requests, 6 per wave)
            const int kb = mac * 4 + j;
            const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(p.B + (BMODE == 2 ? (long long)mac * 384 : (long long)kb * p.bsb)), 0,
                p.skipB ? 0 : (int)(p.rowsB * p.ldb), SRD_FLAGS);
            static_for<6>([&](auto BI) {
                constexpr int bi = decltype(BI)::value;
                const int b = bi * NL + lw;
                int vo;
                if constexpr (BMODE == 2) {                              // a quarter of the rows, 384 bytes each
                    const int s = (j * 24 + b) * 64 + lane, row = s / 24, chunk = s - row * 24;
                    vo = row * (int)p.ldb + chunk * 16;
                } else {
                    const int s = b * 64 + lane, row = s / 6, chunk = s - row * 6;
                    vo = row * (int)p.ldb + chunk * 16;
                }
                dma16x<NTB ? 2 : 0>(sb, lds + ((cnt * NL + lw) & 127) * 1024, vo);
                ++cnt;
                window_wait<W>();
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && p.stamps) {
        long long* st = p.stamps + 2 * blockIdx.x;
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

struct Ctx {
    long long M, R, K, N, lda, ldb, ldy;
    char *xa, *xb, *w, *wblk, *yp3;
    float *z, *stats, *bias, *gamma, *beta;
    int* rowmap;
    long long* stamps;
};

static void report_stamps(Ctx& c, int nwg, double bytes_per_wg) {
    std::vector<long long> h(4096 * 2);
    hipMemcpy(h.data(), c.stamps, 4096 * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz, us;
    for (int i = 0; i < nwg; ++i)
        if (h[2 * i + 1] > 0) { ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); us.push_back((double)h[2 * i + 1] * 0.01); }
    if (ghz.empty()) return;
    std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
    printf("  in-kernel: clock %.3f GHz, K loop median %.1f us (min %.1f max %.1f)", ghz[ghz.size() / 2], us[us.size() / 2], us.front(), us.back());
    if (bytes_per_wg > 0) printf(", median CU intake %.1f GB/s", bytes_per_wg / us[us.size() / 2] * 1e-3);
}

template <typename F>
static double time_launches(F&& launch, int warm, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < warm; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

template <int Q, int BMODE, int W, int NTA, int NTB>
static void run_dma(Ctx& c, const char* what, int skipA, int skipB) {
    DmaArgs a = {};
    a.A = c.xa; a.A2 = c.xb; a.rowmap = c.rowmap; a.M = (int)c.M; a.rowsA = 96; a.KB = (int)((c.K + 15) / 16); a.lda = c.lda;
    a.res_bytes = c.R * c.lda; a.rowsB = 256; a.skipA = skipA; a.skipB = skipB; a.stamps = c.stamps;
    if (BMODE == 1) { a.B = c.wblk; a.ldb = 96; a.bsb = 256 * 96; } else { a.B = c.w; a.ldb = c.ldb; a.bsb = 96; }
    const int nwg = (int)((c.M + 95) / 96);
    static bool conf = false;
    if (!conf) { GTE_SET_LDS((dma_intake_kernel<Q, BMODE, W, NTA, NTB>), 128 * 1024); conf = true; }
    hipMemset(c.stamps, 0, 4096 * 16);
    const double us = time_launches([&] { hipLaunchKernelGGL((dma_intake_kernel<Q, BMODE, W, NTA, NTB>), dim3(nwg), dim3(256), 128 * 1024, nullptr, a); },
                                    400, 100);
    const double kb = 2.0 * a.KB;
    const double bytes = (skipA ? 0.0 : 96.0 * 96 * kb) + (skipB ? 0.0 : 256.0 * 96 * kb);
    printf("requests only  A %3d B/row%s  B %-11s%s  window %2d%s%s: %6.1f us/launch  %5.1f GB/s per CU (launch)", Q == 8 ? 964 : 96 * Q, NTA ? " nt" : "   ",
           BMODE == 0 ? "row-major" : (BMODE == 1 ? "block-major" : "rows x 384"), NTB ? " nt" : "   ", W, skipA ? "  [no A]" : "", skipB ? "  [no B]" : "",
           us, bytes / us * 1e-3);
    report_stamps(c, nwg, bytes);
    printf("   %s\n", what);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "gemm";
    Ctx c = {};
    c.K = 831; c.N = 256;
    const int pages = 1200, prow = 244, bpages = argc > 2 ? atoi(argv[2]) : 100;
    c.R = (long long)pages * prow; c.M = (long long)bpages * prow;
    c.lda = gte_p3_row_bytes(c.K); c.ldb = 2 * c.lda; c.ldy = gte_p3_row_bytes(c.N);
    const bool zero = getenv("ABL_ZERO") != nullptr;
    float* tmp;
    hipMalloc(&tmp, c.R * c.K * 4);
    hipMalloc(&c.xa, (c.R + 1) * c.lda); hipMalloc(&c.xb, (c.R + 1) * c.lda);
    hipMalloc(&c.w, 256 * c.ldb); hipMalloc(&c.wblk, 256 * c.ldb);
    hipMalloc(&c.z, c.M * c.N * 4); hipMalloc(&c.yp3, c.M * c.ldy); hipMalloc(&c.stats, 2 * c.M * 4);
    hipMalloc(&c.bias, 1024); hipMalloc(&c.gamma, 1024); hipMalloc(&c.beta, 1024);
    hipMalloc(&c.rowmap, (c.M + 256) * 4); hipMalloc(&c.stamps, 4096 * 16);
    hipMemset(c.stamps, 0, 4096 * 16);
    if (zero) {
        hipMemset(c.xa, 0, (c.R + 1) * c.lda); hipMemset(c.xb, 0, (c.R + 1) * c.lda); hipMemset(c.w, 0, 256 * c.ldb);
    } else {
        fill_kernel<<<(unsigned)((c.R * c.K + 255) / 256), 256>>>(tmp, c.R * c.K, 1u);
        gte_p3_from_f32(tmp, c.K, c.R, c.K, 0, c.xa, c.lda, nullptr);
        fill_kernel<<<(unsigned)((c.R * c.K + 255) / 256), 256>>>(tmp, c.R * c.K, 7u);
        gte_p3_from_f32(tmp, c.K, c.R, c.K, 0, c.xb, c.lda, nullptr);
        fill_kernel<<<(unsigned)((256 * 2 * c.K + 255) / 256), 256>>>(tmp, 256 * 2 * c.K, 3u);
        // (both K segments of W as one [256][2 x 52 blocks] image: segment 1 at block 52)
        gte_p3_from_f32(tmp, 2 * c.K, 256, c.K, 0, c.w, c.ldb, nullptr);
        gte_p3_from_f32(tmp + c.K, 2 * c.K, 256, c.K, 0, c.w + c.lda, c.ldb, nullptr);
    }
    hipMemset(c.wblk, 0, 256 * c.ldb);
    if (!zero) {                                       // block-major copy of the weights: block fb = a [256][16] image at fb * 256 * 96
        float* tw;
        hipMalloc(&tw, 256 * 2 * c.K * 4);
        fill_kernel<<<(unsigned)((256 * 2 * c.K + 255) / 256), 256>>>(tw, 256 * 2 * c.K, 3u);
        const int kb1 = (int)((c.K + 15) / 16);
        for (int sgm = 0; sgm < 2; ++sgm)
            for (int fb = 0; fb < kb1; ++fb) {
                const long long cols = std::min<long long>(16, c.K - 16 * fb);
                gte_p3_from_f32(tw + sgm * c.K + 16 * fb, 2 * c.K, 256, cols, 0, c.wblk + (long long)(sgm * kb1 + fb) * 256 * 96, 96, nullptr);
            }
        hipDeviceSynchronize();
        hipFree(tw);
    }
    fill_kernel<<<1, 256>>>(c.bias, 256, 11u); fill_kernel<<<1, 256>>>(c.gamma, 256, 12u); fill_kernel<<<1, 256>>>(c.beta, 256, 13u);
    {
        std::vector<int> rm(c.M + 256, (int)c.R);
        unsigned s = 12345u;
        for (int b = 0; b < bpages; ++b) {
            s = s * 1664525u + 1013904223u;
            const int pg = (int)((s >> 8) % pages);
            for (int i = 0; i < prow; ++i) rm[(long long)b * prow + i] = pg * prow + i;
        }
        hipMemcpy(c.rowmap, rm.data(), rm.size() * 4, hipMemcpyHostToDevice);
    }
    hipDeviceSynchronize();
    hipFree(tmp);

    if (!strcmp(mode, "gemm")) {
        const int rows = getenv("ABL_ROWS") ? atoi(getenv("ABL_ROWS")) : 0;
        gte_gemm_p3_set_ln_rows(rows);
        P3Gemm p = {};
        p.A1 = c.xa; p.lda1 = c.lda; p.KB1 = (int)((c.K + 15) / 16);
        p.A2 = c.xb; p.lda2 = c.lda; p.KB2 = p.KB1;
        p.B = c.w; p.ldb = c.ldb; p.C = c.z; p.ldc = c.N; p.bias = c.bias;
        p.bsa1 = p.bsa2 = p.bsb = 96;
        const bool bblk = getenv("ABL_BBLK") != nullptr;
        if (bblk) { p.B = c.wblk; p.ldb = 96; p.bsb = 256 * 96; }
        p.M = (int)c.M; p.N = (int)c.N; p.splits = 1;
        p.ln_gamma = c.gamma; p.ln_beta = c.beta; p.ln_relu = 1;
        p.lnf_y = nullptr; p.lnf_yp3 = c.yp3; p.lnf_ldp = c.ldy; p.lnf_stats = c.stats; p.lnf_eps = 1e-5f;
        p.rowsA = c.rowmap; p.res_bytes = c.R * c.lda; p.rows_both = 1; p.res_bytes2 = c.R * c.lda;
        p.slab = reinterpret_cast<float*>(c.stamps);
        const int bm = lnb_row_tile(c.M);
        const bool sq = getenv("ABL_SQ") != nullptr;
        if (sq) { p.B = c.wblk; p.ldb = 96; p.bsb = 256 * 96; }
        auto launch = [&] {
            if (sq && bm == 96) launch_sq<3, ABL_NL, 4>(p, nullptr);
            else if (sq) launch_sq<4, ABL_NL, 4>(p, nullptr);
            else if (bm == 96) launch_lw_lnb<1, 8, 3, 1, 4, 4>(p, nullptr);
            else launch_lw_lnb<2, 4, 2, 2, 4, 4>(p, nullptr);
        };
        if (getenv("ABL_REF")) {                        // the product of the shipped kernel (row-major weights) against this launch, element by element
            P3Gemm q = p;
            float* zr; hipMalloc(&zr, c.M * c.N * 4);
            q.B = c.w; q.ldb = c.ldb; q.bsb = 96; q.C = zr;
            if (bm == 96) launch_lw_lnb<1, 8, 3, 1, 4, 4>(q, nullptr); else launch_lw_lnb<2, 4, 2, 2, 4, 4>(q, nullptr);
            launch();
            hipDeviceSynchronize();
            std::vector<float> a(c.M * c.N), b(c.M * c.N);
            hipMemcpy(a.data(), zr, a.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), c.z, b.size() * 4, hipMemcpyDeviceToHost);
            long long bad = 0, first = -1; double mx = 0;
            for (size_t i = 0; i < a.size(); ++i)
                if (memcmp(&a[i], &b[i], 4)) { if (first < 0) first = (long long)i; ++bad; mx = std::max(mx, (double)fabsf(a[i] - b[i])); }
            printf("against the shipped kernel: %lld of %zu elements differ (max |d| %.3g), first at row %lld col %lld\n", bad, a.size(), mx,
                   first < 0 ? -1 : first / c.N, first < 0 ? -1 : first % c.N);
            hipFree(zr);
        }
        const double us = time_launches(launch, 3000, 200);
        const double gf = 2.0 * c.M * c.N * 2 * c.K * 1e-9;
        printf("L0 forward %lld x %lld x %lld, %d-row tiles, abl %3d%s%s: %6.1f us  %5.1f TF fp32-eq = %.3f of 416.7", c.M, c.N, 2 * c.K, bm, P3_ABL,
               zero ? " zero operands" : "", sq ? " SQ kernel (weights to registers, 64-deep A slots)" : (bblk ? " block-major weights" : ""), us, gf / us * 1e3, gf / us * 1e3 / 416.7);
        if (getenv("ABL_CHECK")) {                        // checksum of z and of the image (the two weight layouts must agree bit for bit)
            std::vector<float> hz(c.M * c.N);
            std::vector<unsigned char> hy(c.M * c.ldy);
            hipMemcpy(hz.data(), c.z, hz.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hy.data(), c.yp3, hy.size(), hipMemcpyDeviceToHost);
            unsigned long long h1 = 1469598103934665603ull, h2 = h1;
            for (float v : hz) { unsigned u; memcpy(&u, &v, 4); h1 = (h1 ^ u) * 1099511628211ull; }
            for (unsigned char v : hy) h2 = (h2 ^ v) * 1099511628211ull;
            printf("  z %016llx y %016llx", h1, h2);
        }
#if P3_ABL & 16
        report_stamps(c, (int)((c.M + bm - 1) / bm), (bm + 256.0) * 96 * 2 * p.KB1);
#endif
        printf("\n");
        return 0;
    }
    // ---- requests only ----
    run_dma<1, 0, 18, 0, 0>(c, "the kernel's fetch shape (two 36 KB stages in flight)", 0, 0);
    run_dma<1, 0, 18, 0, 0>(c, "", 1, 0);
    run_dma<1, 0, 18, 0, 0>(c, "", 0, 1);
    run_dma<1, 0, 36, 0, 0>(c, "", 0, 0);
    run_dma<1, 0, 54, 0, 0>(c, "", 0, 0);
    run_dma<1, 0, 18, 1, 0>(c, "", 0, 0);
    run_dma<1, 1, 18, 0, 0>(c, "weights block-major", 0, 0);
    run_dma<1, 1, 18, 0, 0>(c, "", 1, 0);
    run_dma<1, 1, 36, 0, 0>(c, "", 0, 0);
    run_dma<1, 1, 18, 1, 0>(c, "", 0, 0);
    run_dma<1, 2, 18, 0, 0>(c, "weights in 384-byte runs", 0, 0);
    run_dma<1, 2, 18, 0, 0>(c, "", 1, 0);
    run_dma<2, 0, 18, 0, 0>(c, "A in 192-byte runs", 0, 1);
    run_dma<4, 0, 18, 0, 0>(c, "A in 384-byte runs", 0, 1);
    run_dma<2, 1, 18, 0, 0>(c, "", 0, 0);
    run_dma<4, 1, 18, 0, 0>(c, "", 0, 0);
    run_dma<4, 1, 36, 0, 0>(c, "", 0, 0);
    run_dma<4, 1, 18, 1, 0>(c, "", 0, 0);
    run_dma<4, 1, 36, 1, 0>(c, "", 0, 0);
    run_dma<4, 2, 18, 0, 0>(c, "", 0, 0);
    run_dma<4, 2, 36, 1, 0>(c, "", 0, 0);
    run_dma<8, 1, 18, 0, 0>(c, "A in 96-byte pieces, the four blocks of a row group back to back from one wave (964 = 96 x 4)", 0, 0);
    run_dma<8, 1, 18, 0, 0>(c, "", 0, 1);
    run_dma<4, 1, 18, 0, 0>(c, "", 1, 0);
    return 0;
}
