// Micro-benchmark (not shipped): PERSISTENT form of the software-pipelined fp32 MFMA GEMM (gemm_pipe.hip), NT layout.
//   C[M,N] = A[M,K] * B[N,K]^T.  A workgroup walks the tiles u = first, first + G, first + 2G, ... of a grid of G workgroups;
//   the load cursor runs on into the workgroup's NEXT tile, so that tile's first two K stages are requested under the current
//   tile's last two stages (no prologue latency after the first tile, no workgroup dispatch between tiles); the C stores of
//   a tile are issued and NOT waited for (they drain under the next tile's first stages).
// Why: gemm_rounds.py / gemm_pipe16 (round 2): the K loop runs at 96 % of the clock-limited rate at K = 4096, but every tile
// pays ~6 us of start-up (dispatch, two memory round trips before the first MFMA) and ~2 us of stores: 12-25 % at K = 512-831.
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 gemm_persist.hip -o gemm_persist ; run: ./gemm_persist [M N K]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned SRD_FLAGS = 0x00020000u;
constexpr int BK = 32, KPAD = BK + 4;

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int M_MFMA = 0x008, M_VMEM_R = 0x020, M_DS_R = 0x100, M_DS_W = 0x200, M_VALU = 0x002, M_SALU = 0x004;

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned n) {
    const unsigned xcd = bid % 8, q = n / 8, r = n % 8;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / 8;
}

template <int BM, int BN, int WM, int WN, bool PERSIST>
__global__ void __launch_bounds__(256)
gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc, int M, int N,
        int K, unsigned long long* __restrict__ stamps) {
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c_start = __builtin_amdgcn_s_memtime();
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NCA = BM * (BK / 4) / 256, NCB = BN * (BK / 4) / 256;
    constexpr int BUF = (BM + BN) * KPAD;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int ntile = tiles_m * tiles_n;
    const int G = (int)gridDim.x;
    const int first = (int)xcd_remap(blockIdx.x, gridDim.x);   // workgroups of one XCD hold consecutive tiles at any time
    const int T = (K + BK - 1) / BK;

    const int kq = tid & 7, r0 = tid >> 3;
    int voffA[NCA], voffB[NCB];
#pragma unroll
    for (int i = 0; i < NCA; ++i) voffA[i] = ((r0 + 32 * i) * lda + kq * 4) * 4;
#pragma unroll
    for (int i = 0; i < NCB; ++i) voffB[i] = ((r0 + 32 * i) * ldb + kq * 4) * 4;
    const int wofs = r0 * KPAD + kq * 4;

    // ---- load cursor: (unit, K tile) of the next stage to describe; everything block-uniform (scalar) ----
    struct Desc { __amdgpu_buffer_rsrc_t sa, sb; int oob, vc; };
    int cu = first, ct = 0;
    const float* cbA = nullptr; const float* cbB = nullptr;
    int cbytesA = 0, cbytesB = 0;
    auto enter_unit = [&]() {
        const bool ok = cu < ntile;
        const int u = ok ? cu : 0;
        const int m0 = (u / tiles_n) * BM, n0 = (u % tiles_n) * BN;
        const int rowsA = min(BM, M - m0), rowsB = min(BN, N - n0);
        cbA = A + (size_t)m0 * lda;
        cbB = B + (size_t)n0 * ldb;
        cbytesA = ok ? ((rowsA - 1) * lda + K) * 4 : 0;
        cbytesB = ok ? ((rowsB - 1) * ldb + K) * 4 : 0;
    };
    enter_unit();
    auto describe = [&]() {
        Desc d;
        const int kl = ct * BK;
        const int remA = cbytesA - kl * 4, remB = cbytesB - kl * 4;
        d.sa = __builtin_amdgcn_make_buffer_rsrc((void*)(cbA + kl), 0, remA > 0 ? remA : 0, SRD_FLAGS);
        d.sb = __builtin_amdgcn_make_buffer_rsrc((void*)(cbB + kl), 0, remB > 0 ? remB : 0, SRD_FLAGS);
        d.vc = K - kl - kq * 4;
        d.oob = ((d.vc - 1) >> 31) & (int)0x80000000;
        ++ct;
        if (ct == T) { ct = 0; cu += PERSIST ? G : ntile; enter_unit(); }      // uniform; once per tile
        return d;
    };

    f32x4 ra[NCA], rb[NCB];
    int vc_st = 0;
    Desc D;
    auto issue_loads_into = [&](f32x4 (&xa)[NCA], f32x4 (&xb)[NCB], const Desc& d, auto FROM, auto TO) {
#pragma unroll
        for (int j = decltype(FROM)::value; j < decltype(TO)::value; ++j) {
            if (j < NCA) xa[j < NCA ? j : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.sa, voffA[j < NCA ? j : 0] | d.oob, 0, 0));
            else xb[j >= NCA ? j - NCA : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d.sb, voffB[j >= NCA ? j - NCA : 0] | d.oob, 0, 0));
        }
    };
    auto store_chunks_from = [&](const f32x4 (&xa)[NCA], const f32x4 (&xb)[NCB], int vc, int buf, auto FROM, auto TO) {
        float* wa = lds + buf * BUF + wofs;
        float* wb = wa + BM * KPAD;
#pragma unroll
        for (int j = decltype(FROM)::value; j < decltype(TO)::value; ++j) {
            f32x4 v = j < NCA ? xa[j < NCA ? j : 0] : xb[j >= NCA ? j - NCA : 0];
            v.y = vc > 1 ? v.y : 0.f;
            v.z = vc > 2 ? v.z : 0.f;
            v.w = vc > 3 ? v.w : 0.f;
            if (j < NCA) *reinterpret_cast<f32x4*>(wa + j * 32 * KPAD) = v;
            else *reinterpret_cast<f32x4*>(wb + (j - NCA) * 32 * KPAD) = v;
        }
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();
    f32x4 fa[2][TM], fb[2][TN];
    const int frA = ((wm * TM) * 32 + (lane & 31)) * KPAD + (lane >> 5) * 4;
    const int frB = BM * KPAD + ((wn * TN) * 32 + (lane & 31)) * KPAD + (lane >> 5) * 4;
    auto read_frags = [&](int buf, int kg, int slot) {
        const float* pa = lds + buf * BUF + frA + kg * 8;
        const float* pb = lds + buf * BUF + frB + kg * 8;
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[slot][a] = *reinterpret_cast<const f32x4*>(pa + a * 32 * KPAD);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[slot][b] = *reinterpret_cast<const f32x4*>(pb + b * 32 * KPAD);
    };
    auto mfma_group = [&](int slot) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][a][tt], fb[slot][b][tt], acc[a][b], 0, 0, 0);
    };
    constexpr int NC = NCA + NCB, H = NC / 2;
    constexpr int NMF = 4 * TM * TN;
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, H>;
    using IN = std::integral_constant<int, NC>;

    // prologue (once per workgroup): stage 0 -> LDS[0] through a second register set, stage 1 -> staging registers
    {
        f32x4 pa[NCA], pb[NCB];
        const Desc D0 = describe();
        issue_loads_into(pa, pb, D0, I0{}, IN{});
        D = describe();
        issue_loads_into(ra, rb, D, I0{}, IN{});
        store_chunks_from(pa, pb, D0.vc, 0, I0{}, IN{});
        vc_st = D.vc;
        D = describe();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    read_frags(0, 0, 0);
    int cur = 0;
    const int col_l = lane & 31, hrow = (lane >> 5) * 4;

    for (int u = first; u < ntile; u += PERSIST ? G : ntile) {
        for (int t = 0; t < T; ++t, cur ^= 1) {
            read_frags(cur, 1, 1);
            store_chunks_from(ra, rb, vc_st, cur ^ 1, I0{}, IH{});
            issue_loads_into(ra, rb, D, I0{}, IH{});
            mfma_group(0);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
#pragma unroll
            for (int i = 0; i < H; ++i) { SGB(M_MFMA, 1); SGB(M_VALU, 3); SGB(M_DS_W, 1); SGB(M_MFMA, 1); SGB(M_VMEM_R, 1); }
            SGB(M_MFMA, NMF - (TM + TN) - 2 * H);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(cur, 2, 0);
            store_chunks_from(ra, rb, vc_st, cur ^ 1, IH{}, IN{});
            issue_loads_into(ra, rb, D, IH{}, IN{});
            mfma_group(1);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
#pragma unroll
            for (int i = 0; i < NC - H; ++i) { SGB(M_MFMA, 1); SGB(M_VALU, 3); SGB(M_DS_W, 1); SGB(M_MFMA, 1); SGB(M_VMEM_R, 1); }
            SGB(M_MFMA, NMF - (TM + TN) - 2 * (NC - H));
            __builtin_amdgcn_sched_barrier(0);
            vc_st = D.vc;
            read_frags(cur, 3, 1);
            D = describe();
            mfma_group(0);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
#pragma unroll
            for (int i = 0; i < NMF - (TM + TN); ++i) { SGB(M_MFMA, 1); SGB(M_SALU, 6); SGB(M_VALU, 1); }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            read_frags(cur ^ 1, 0, 0);
            mfma_group(1);
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) { SGB(M_MFMA, 1); SGB(M_DS_R, 1); }
            SGB(M_MFMA, NMF - (TM + TN));
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue of unit u: stores are issued and left in flight
        const int m0 = (u / tiles_n) * BM, n0 = (u % tiles_n) * BN;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + (wn * TN + b) * 32 + col_l;
            if (col < N) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int rbase = m0 + (wm * TM + a) * 32 + hrow;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < M) C[(size_t)row * ldc + col] = acc[a][b][r];
                    }
                }
            }
        }
        zero_acc();
    }
    if (stamps && threadIdx.x == 0) {
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        stamps[blockIdx.x * 4 + 0] = t_start; stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[blockIdx.x * 4 + 2] = xcc; stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - c_start; (void)hwid;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int BM, int BN, int WM, int WN, bool PERSIST>
void run(const char* name, int wgs_per_cu, const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
         const std::vector<float>& hB) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int grid = PERSIST ? (tiles < 256 * wgs_per_cu ? tiles : 256 * wgs_per_cu) : tiles;
    const size_t shm = 2 * (BM + BN) * KPAD * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_nt<BM, BN, WM, WN, PERSIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const int lda_t = getenv("ALIAS_A") ? 0 : K;            // ALIAS_A=1: every A row reads row 0 (always cache hits; results wrong, timing only)
    auto launch = [&]() { hipLaunchKernelGGL((gemm_nt<BM, BN, WM, WN, PERSIST>), dim3(grid), dim3(256), shm, 0, dA, lda_t, dB, K, dC, N, M, N, K, (unsigned long long*)nullptr); };
    CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int s = 0; s < 6000; ++s) {
        const int m = (s * 7919 + (s % 3 ? M - 1 - s % 200 : 0)) % M, n = (s * 104729) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
        if (!getenv("ALIAS_A") && (double)hC[(size_t)m * N + n] != ref) { if (bad < 5) printf("  mismatch (%d,%d): %f vs %f\n", m, n, hC[(size_t)m * N + n], ref); ++bad; }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e0));
    const int reps = 40;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-30s M=%d N=%d K=%d tiles=%d grid=%d  %8.1f us  %6.1f TF  %s\n", name, M, N, K, tiles, grid, ms * 1e3,
           2.0 * M * N * K / ms / 1e9, bad ? "MISMATCH" : "exact");
    {   // per-workgroup wall-clock stamps (100 MHz): when do workgroups start and end?
        unsigned long long* dS; CK(hipMalloc(&dS, (size_t)grid * 32));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_nt<BM, BN, WM, WN, PERSIST>), dim3(grid), dim3(256), shm, 0, dA, K, dB, K, dC, N, M, N, K, dS);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hS((size_t)grid * 4);
        CK(hipMemcpy(hS.data(), dS, hS.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < grid; ++i) { if (hS[i * 4] < t0) t0 = hS[i * 4]; if (hS[i * 4 + 1] > t1) t1 = hS[i * 4 + 1]; }
        std::vector<double> dur, st, en;
        for (int i = 0; i < grid; ++i) { dur.push_back((hS[i * 4 + 1] - hS[i * 4]) * 0.01); st.push_back((hS[i * 4] - t0) * 0.01); en.push_back((hS[i * 4 + 1] - t0) * 0.01); }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        printf("    span %.1f us | WG duration p5/p50/p95/max %.1f/%.1f/%.1f/%.1f | start p50/p95/max %.1f/%.1f/%.1f | end p5/p50/p95/max %.1f/%.1f/%.1f/%.1f\n",
               (t1 - t0) * 0.01, pct(dur, .05), pct(dur, .5), pct(dur, .95), pct(dur, 1.0), pct(st, .5), pct(st, .95), pct(st, 1.0),
               pct(en, .05), pct(en, .5), pct(en, .95), pct(en, 1.0));
        {
            double ck = 0; int nck = 0;
            for (int i = 0; i < grid; ++i) { const double us = dur[i]; if (us > 5) { ck += (double)hS[i * 4 + 3] / us; ++nck; } }
            printf("    in-kernel shader clock (s_memtime / s_memrealtime): %.0f MHz -> fp32 MFMA peak at that clock %.1f TF\n", ck / nck, ck / nck * 1e6 * 256 * 256 / 1e12);
        }
        if (false) {
            double sum[8] = {0}, mx[8] = {0}; int cnt[8] = {0};
            std::vector<double> cu_end(8 * 64, 0.0); std::vector<int> cu_n(8 * 64, 0);
            for (int i = 0; i < grid; ++i) {
                if (st[i] > 5.0) continue;
                const int x = (int)(hS[i * 4 + 2] & 7);
                const unsigned hw = (unsigned)hS[i * 4 + 3];
                const int cu = (int)((hw >> 8) & 0xf), sh = (int)((hw >> 12) & 1), se = (int)((hw >> 13) & 7);   // HW_ID: cu_id[11:8], sh_id[12], se_id[15:13]
                const int key = x * 64 + (se * 2 + sh) * 16 + cu;
                sum[x] += dur[i]; if (dur[i] > mx[x]) mx[x] = dur[i]; ++cnt[x];
                if (key < 512) { cu_end[key] = std::max(cu_end[key], en[i]); ++cu_n[key]; }
            }
            printf("    first round by XCD (n, mean, max us):");
            for (int x = 0; x < 8; ++x) printf(" [%d: %d %.1f %.1f]", x, cnt[x], cnt[x] ? sum[x] / cnt[x] : 0.0, mx[x]);
            int hist[8] = {0};
            for (int k = 0; k < 512; ++k) if (cu_n[k] >= 0 && cu_n[k] < 8) ++hist[cu_n[k]];
            printf("\n    first-round workgroups per (XCD, SE, SH, CU) slot -> number of slots: 0:%d 1:%d 2:%d 3:%d 4:%d\n", hist[0], hist[1], hist[2], hist[3], hist[4]);
        }
        CK(hipFree(dS));
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 24495, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 831;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 28) - 8); }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 29) - 4); }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<128, 128, 2, 2, false>("128x128 one tile per WG", 0, dA, dB, dC, M, N, K, hA, hB);
        run<128, 128, 2, 2, true>("128x128 persistent 1 WG/CU", 1, dA, dB, dC, M, N, K, hA, hB);
        run<128, 128, 2, 2, true>("128x128 persistent 2 WG/CU", 2, dA, dB, dC, M, N, K, hA, hB);
        run<64, 128, 2, 2, false>("64x128 one tile per WG", 0, dA, dB, dC, M, N, K, hA, hB);
        run<64, 128, 2, 2, true>("64x128 persistent 2 WG/CU", 2, dA, dB, dC, M, N, K, hA, hB);
        run<64, 128, 2, 2, true>("64x128 persistent 3 WG/CU", 3, dA, dB, dC, M, N, K, hA, hB);
    }
    return 0;
}
