// Micro-benchmark (not shipped): fp32 GEMM on the bf16 matrix pipe through a three-way operand split.
//   C[M,N] = A[M,K] * B[N,K]^T, fp32 in, fp32 out.  Every fp32 operand x is cut into three bf16 pieces
//   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even; |m| <= 2^-8 |x|, |l| <= 2^-16 |x|,
//   h + m + l == x up to 2^-24 |x|) when the tile goes registers -> LDS; the piece products are exact in fp32 and
//   are accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//     NPROD = 6: hh + hm + mh + mm + hl + lh          (dropped: ml + lm + ll <= 2^-23 |a b|)
//     NPROD = 8: + ml + lm                            (dropped: ll <= 2^-32 |a b|)
//     NPROD = 3: hh + hm + mh                         (the usual "bf16x3": ~2^-16, for comparison only)
//   One 32x32x16 bf16 MFMA is 32 cycles for K = 16; the fp32 32x32x2 MFMA is 64 cycles for K = 2: six products cost
//   192 cycles per K = 16 against 512.
// build: hipcc -O3 --offload-arch=gfx950 profiles/micro/gemm_split.hip -o profiles/micro/gemm_split
// run:   profiles/micro/gemm_split [M N K]         (K is padded to a multiple of 96 with zeros)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int BM = 128, BN = 128, BK = 32, LDP = BK + 8;          // LDS row stride in bf16 (80 bytes)
constexpr int PLANE = 128 * LDP;                                  // bf16 elements per piece plane

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned xcd = bid % 8, q = nblocks / 8, r = nblocks % 8;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / 8;
}

struct Pieces { bf16x8 h, m, l; };

__device__ __forceinline__ Pieces split8(const f32x4 a, const f32x4 b) {
    Pieces p;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? a[j] : b[j - 4];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        p.h[j] = h; p.m[j] = m; p.l[j] = (__bf16)r2;
    }
    return p;
}

template <int NPROD>
__global__ void __launch_bounds__(256, 2)
gemm_split_nt(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
              int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) u16 sA[3 * PLANE];
    __shared__ __attribute__((aligned(16))) u16 sB[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = (int)(blockIdx.x / tiles_n) * BM, n0 = (int)(blockIdx.x % tiles_n) * BN;

    const int pr = tid >> 2, pc = (tid & 3) * 8;
    f32x4 ra[2][2], rb[2][2];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ar = min(m0 + pr + 64 * h, M - 1), br = min(n0 + pr + 64 * h, N - 1);
            const f32x4* pa = reinterpret_cast<const f32x4*>(A + (size_t)ar * lda + k0 + pc);
            const f32x4* pb = reinterpret_cast<const f32x4*>(B + (size_t)br * ldb + k0 + pc);
            ra[h][0] = pa[0]; ra[h][1] = pa[1];
            rb[h][0] = pb[0]; rb[h][1] = pb[1];
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const Pieces a = split8(ra[h][0], ra[h][1]), b = split8(rb[h][0], rb[h][1]);
            const int o = (pr + 64 * h) * LDP + pc;
            *reinterpret_cast<bf16x8*>(&sA[o]) = a.h;
            *reinterpret_cast<bf16x8*>(&sA[PLANE + o]) = a.m;
            *reinterpret_cast<bf16x8*>(&sA[2 * PLANE + o]) = a.l;
            *reinterpret_cast<bf16x8*>(&sB[o]) = b.h;
            *reinterpret_cast<bf16x8*>(&sB[PLANE + o]) = b.m;
            *reinterpret_cast<bf16x8*>(&sB[2 * PLANE + o]) = b.l;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_stage(0);
    store_stage();
    __syncthreads();
    const int i32 = lane & 31, kh = (lane >> 5) * 8;
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (k0 + BK < K) load_stage(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            bf16x8 fa[3][2], fb[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    fa[p][a] = *reinterpret_cast<const bf16x8*>(&sA[p * PLANE + ((wm * 2 + a) * 32 + i32) * LDP + ks + kh]);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    fb[p][b] = *reinterpret_cast<const bf16x8*>(&sB[p * PLANE + ((wn * 2 + b) * 32 + i32) * LDP + ks + kh]);
            }
            // smallest terms first
            constexpr int PA[8] = {2, 1, 2, 0, 1, 1, 0, 0};      // ll is never formed
            constexpr int PB[8] = {1, 2, 0, 2, 1, 0, 1, 0};
            constexpr int FIRST = NPROD == 8 ? 0 : (NPROD == 6 ? 2 : 5);
#pragma unroll
            for (int q = FIRST; q < 8; ++q)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (k0 + BK < K) store_stage();
        __syncthreads();
    }
    const int hrow = (lane >> 5) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i32;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + a) * 32 + hrow + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * ldc + col] = acc[a][b][r];
            }
        }
}


// ---- v2: BK = 16, LDS double buffer, one barrier per stage, packed split arithmetic -------------------------------
constexpr int BK2 = 16, LDP2 = BK2 + 8, PLANE2 = 128 * LDP2;      // 48-byte rows: conflict-free ds_read_b128 / ds_write_b128
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> packed bf16 pieces (h, m, l), 9 VALU: cvt_pk, shl, and, pk_add, cvt_pk, shl, and, pk_add, cvt_pk
__device__ __forceinline__ void split2(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    f32x2 hf;
    hf.x = __builtin_bit_cast(float, h << 16);
    hf.y = __builtin_bit_cast(float, h & 0xffff0000u);
    const f32x2 r1 = x - hf;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    f32x2 mf;
    mf.x = __builtin_bit_cast(float, m << 16);
    mf.y = __builtin_bit_cast(float, m & 0xffff0000u);
    const f32x2 r2 = r1 - mf;
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

struct Pk8 { uint4 h, m, l; };
__device__ __forceinline__ Pk8 split8p(const f32x4 a, const f32x4 b) {
    Pk8 p;
    split2(f32x2{a.x, a.y}, p.h.x, p.m.x, p.l.x);
    split2(f32x2{a.z, a.w}, p.h.y, p.m.y, p.l.y);
    split2(f32x2{b.x, b.y}, p.h.z, p.m.z, p.l.z);
    split2(f32x2{b.z, b.w}, p.h.w, p.m.w, p.l.w);
    return p;
}

template <int NPROD>
__global__ void __launch_bounds__(256, 2)
gemm_split_nt2(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
               int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) u16 sm[2 * 6 * PLANE2];              // [buf][A h m l | B h m l]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = (int)(blockIdx.x / tiles_n) * BM, n0 = (int)(blockIdx.x % tiles_n) * BN;

    const int pr = tid >> 1, pc = (tid & 1) * 8;                                 // row, first k of this thread's 8
    const float* pa = A + (size_t)min(m0 + pr, M - 1) * lda + pc;
    const float* pb = B + (size_t)min(n0 + pr, N - 1) * ldb + pc;
    f32x4 ra[2], rb[2];
    auto load_stage = [&](int k0) {
        ra[0] = *reinterpret_cast<const f32x4*>(pa + k0); ra[1] = *reinterpret_cast<const f32x4*>(pa + k0 + 4);
        rb[0] = *reinterpret_cast<const f32x4*>(pb + k0); rb[1] = *reinterpret_cast<const f32x4*>(pb + k0 + 4);
    };
    const int wo = pr * LDP2 + pc;
    auto store_stage = [&](int buf) {
        u16* s = sm + buf * 6 * PLANE2;
        const Pk8 a = split8p(ra[0], ra[1]);
        *reinterpret_cast<uint4*>(&s[wo]) = a.h;
        *reinterpret_cast<uint4*>(&s[PLANE2 + wo]) = a.m;
        *reinterpret_cast<uint4*>(&s[2 * PLANE2 + wo]) = a.l;
        const Pk8 b = split8p(rb[0], rb[1]);
        *reinterpret_cast<uint4*>(&s[3 * PLANE2 + wo]) = b.h;
        *reinterpret_cast<uint4*>(&s[4 * PLANE2 + wo]) = b.m;
        *reinterpret_cast<uint4*>(&s[5 * PLANE2 + wo]) = b.l;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nst = K / BK2;
    load_stage(0);
    store_stage(0);
    if (nst > 1) load_stage(BK2);
    __syncthreads();
    const int i32 = lane & 31, kh = (lane >> 5) * 8;
    const int fa_o = (wm * 64 + i32) * LDP2 + kh, fb_o = 3 * PLANE2 + (wn * 64 + i32) * LDP2 + kh;
    for (int t = 0; t < nst; ++t) {
        const u16* s = sm + (t & 1) * 6 * PLANE2;
        bf16x8 fa[3][2], fb[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[p][a] = *reinterpret_cast<const bf16x8*>(&s[p * PLANE2 + fa_o + a * 32 * LDP2]);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[p][b] = *reinterpret_cast<const bf16x8*>(&s[p * PLANE2 + fb_o + b * 32 * LDP2]);
        }
        if (t + 1 < nst) store_stage((t + 1) & 1);           // registers hold tile t+1
        if (t + 2 < nst) load_stage((t + 2) * BK2);
        constexpr int PA[8] = {2, 1, 2, 0, 1, 1, 0, 0};
        constexpr int PB[8] = {1, 2, 0, 2, 1, 0, 1, 0};
        constexpr int FIRST = NPROD == 8 ? 0 : (NPROD == 6 ? 2 : 5);
#pragma unroll
        for (int q = FIRST; q < 8; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);
        __syncthreads();
    }
    const int hrow = (lane >> 5) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i32;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + a) * 32 + hrow + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * ldc + col] = acc[a][b][r];
            }
        }
}


// ---- v3: v2 + branch-free loop, MFMAs interleaved with the split / LDS writes / global loads, the barrier two products
// before the end of the stage so the next stage's fragment reads run under the last MFMAs --------------------------------
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int M_MFMA = 0x008, M_VMEM_R = 0x020, M_DS_R = 0x100, M_DS_W = 0x200, M_VALU = 0x002, M_SALU = 0x004;

struct Frags { bf16x8 a[3][2], b[3][2]; };

template <int NPROD, int PIN>
__global__ void __launch_bounds__(256, 2)
gemm_split_nt3(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
               int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) u16 sm[2 * 6 * PLANE2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    const int m0 = (int)(blockIdx.x / tiles_n) * BM, n0 = (int)(blockIdx.x % tiles_n) * BN;

    const int pr = tid >> 1, pc = (tid & 1) * 8;
    const float* pa = A + (size_t)min(m0 + pr, M - 1) * lda + pc;
    const float* pb = B + (size_t)min(n0 + pr, N - 1) * ldb + pc;
    f32x4 ra[2], rb[2];
    const int klast = K - BK2;
    auto load_stage = [&](int k0) {
        k0 = min(k0, klast);                                  // past the end: the last tile again (never multiplied)
        ra[0] = *reinterpret_cast<const f32x4*>(pa + k0); ra[1] = *reinterpret_cast<const f32x4*>(pa + k0 + 4);
        rb[0] = *reinterpret_cast<const f32x4*>(pb + k0); rb[1] = *reinterpret_cast<const f32x4*>(pb + k0 + 4);
    };
    const int wo = pr * LDP2 + pc;
    auto store_stage = [&](u16* s) {
        const Pk8 a = split8p(ra[0], ra[1]);
        *reinterpret_cast<uint4*>(&s[wo]) = a.h;
        *reinterpret_cast<uint4*>(&s[PLANE2 + wo]) = a.m;
        *reinterpret_cast<uint4*>(&s[2 * PLANE2 + wo]) = a.l;
        const Pk8 b = split8p(rb[0], rb[1]);
        *reinterpret_cast<uint4*>(&s[3 * PLANE2 + wo]) = b.h;
        *reinterpret_cast<uint4*>(&s[4 * PLANE2 + wo]) = b.m;
        *reinterpret_cast<uint4*>(&s[5 * PLANE2 + wo]) = b.l;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int i32 = lane & 31, kh = (lane >> 5) * 8;
    const int fa_o = (wm * 64 + i32) * LDP2 + kh, fb_o = 3 * PLANE2 + (wn * 64 + i32) * LDP2 + kh;
    auto read_frags = [&](Frags& f, const u16* s) {
#pragma unroll
        for (int p = 2; p >= 0; --p) {                        // l planes first: the first products need them
#pragma unroll
            for (int a = 0; a < 2; ++a) f.a[p][a] = *reinterpret_cast<const bf16x8*>(&s[p * PLANE2 + fa_o + a * 32 * LDP2]);
#pragma unroll
            for (int b = 0; b < 2; ++b) f.b[p][b] = *reinterpret_cast<const bf16x8*>(&s[p * PLANE2 + fb_o + b * 32 * LDP2]);
        }
    };
    constexpr int PA[8] = {2, 1, 0, 2, 1, 0, 1, 0};           // a piece / b piece per product, smallest terms first;
    constexpr int PB[8] = {1, 2, 2, 0, 1, 1, 0, 0};           // the last two (mh, hh) need only a.m, a.h, b.h
    constexpr int FIRST = NPROD == 8 ? 0 : (NPROD == 6 ? 2 : 5);
    auto products = [&](const Frags& f, auto Q0, auto Q1) {
#pragma unroll
        for (int q = decltype(Q0)::value; q < decltype(Q1)::value; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[q]][a], f.b[PB[q]][b], acc[a][b], 0, 0, 0);
    };
    using std::integral_constant;
    constexpr int NQ1 = 6 - FIRST;                            // products before the barrier
    // one stage: multiply tile t out of `cur`; tile t+1 (in the staging registers) goes to `nxt`; tile t+2 is loaded
    auto stage = [&](Frags& f, Frags& fn, u16* cur, u16* nxt, int t) {
        store_stage(nxt);
        load_stage((t + 2) * BK2);
        products(f, integral_constant<int, FIRST>{}, integral_constant<int, 6>{});
        if constexpr (PIN) {
#pragma unroll
            for (int i = 0; i < NQ1 * 4; ++i) {
                SGB(M_MFMA, 1);
                SGB(M_VALU, (80 + NQ1 * 4 - 1) / (NQ1 * 4));
                SGB(M_DS_W, 1);
                SGB(M_VMEM_R, 1);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_s_barrier();
        read_frags(fn, nxt);
        products(f, integral_constant<int, 6>{}, integral_constant<int, 8>{});
        if constexpr (PIN) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                SGB(M_MFMA, 1);
                SGB(M_DS_R, 2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nst = K / BK2;                                  // even (K padded to 32)
    u16* s0 = sm;
    u16* s1 = sm + 6 * PLANE2;
    load_stage(0);
    store_stage(s0);
    load_stage(BK2);
    __syncthreads();
    Frags f0, f1;
    read_frags(f0, s0);
    for (int t = 0; t < nst; t += 2) {
        stage(f0, f1, s0, s1, t);
        stage(f1, f0, s1, s0, t + 1);
    }
    const int hrow = (lane >> 5) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i32;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + a) * 32 + hrow + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * ldc + col] = acc[a][b][r];
            }
        }
}


// ---- v4: v3 with NSET staging register sets: the loads of tile t + NSET are issued at the START of stage t ----------
template <int NPROD, int PIN, int NSET, int ABL = 0>   // ABL (timing only): 1 no split, 2 no LDS writes, 4 no barrier, 8 no global loads, 16 no MFMA
__global__ void __launch_bounds__(256, 2)
gemm_split_nt4(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
               int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) u16 sm[2 * 6 * PLANE2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);    // tiles sharing an A row panel on one XCD (one L2)
    const int m0 = (int)(lb / tiles_n) * BM, n0 = (int)(lb % tiles_n) * BN;

    const int pr = tid >> 1, pc = (tid & 1) * 8;
    const float* pa = A + (size_t)min(m0 + pr, M - 1) * lda + pc;
    const float* pb = B + (size_t)min(n0 + pr, N - 1) * ldb + pc;
    f32x4 rs[NSET][4];
    const int klast = K - BK2;
    auto load_stage = [&](auto S, int k0) {
        constexpr int st = decltype(S)::value;
        if constexpr (ABL & 8) { asm volatile("" : "+v"(rs[st][0]), "+v"(rs[st][1]), "+v"(rs[st][2]), "+v"(rs[st][3])); return; }
        k0 = min(k0, klast);                                  // past the end: the last tile again (never multiplied)
        rs[st][0] = *reinterpret_cast<const f32x4*>(pa + k0); rs[st][1] = *reinterpret_cast<const f32x4*>(pa + k0 + 4);
        rs[st][2] = *reinterpret_cast<const f32x4*>(pb + k0); rs[st][3] = *reinterpret_cast<const f32x4*>(pb + k0 + 4);
    };
    const int wo = pr * LDP2 + pc;
    auto store_stage = [&](auto S, u16* s) {
        constexpr int st = decltype(S)::value;
        if constexpr (ABL & 2) { asm volatile("" :: "v"(rs[st][0]), "v"(rs[st][1]), "v"(rs[st][2]), "v"(rs[st][3])); return; }
        if constexpr (ABL & 1) {
            const uint4 a0 = __builtin_bit_cast(uint4, rs[st][0]), a1 = __builtin_bit_cast(uint4, rs[st][1]);
            const uint4 b0 = __builtin_bit_cast(uint4, rs[st][2]), b1 = __builtin_bit_cast(uint4, rs[st][3]);
            *reinterpret_cast<uint4*>(&s[wo]) = a0; *reinterpret_cast<uint4*>(&s[PLANE2 + wo]) = a1;
            *reinterpret_cast<uint4*>(&s[2 * PLANE2 + wo]) = a0; *reinterpret_cast<uint4*>(&s[3 * PLANE2 + wo]) = b0;
            *reinterpret_cast<uint4*>(&s[4 * PLANE2 + wo]) = b1; *reinterpret_cast<uint4*>(&s[5 * PLANE2 + wo]) = b0;
            return;
        }
        const Pk8 a = split8p(rs[st][0], rs[st][1]);
        *reinterpret_cast<uint4*>(&s[wo]) = a.h;
        *reinterpret_cast<uint4*>(&s[PLANE2 + wo]) = a.m;
        *reinterpret_cast<uint4*>(&s[2 * PLANE2 + wo]) = a.l;
        const Pk8 b = split8p(rs[st][2], rs[st][3]);
        *reinterpret_cast<uint4*>(&s[3 * PLANE2 + wo]) = b.h;
        *reinterpret_cast<uint4*>(&s[4 * PLANE2 + wo]) = b.m;
        *reinterpret_cast<uint4*>(&s[5 * PLANE2 + wo]) = b.l;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int i32 = lane & 31, kh = (lane >> 5) * 8;
    const int fa_o = (wm * 64 + i32) * LDP2 + kh, fb_o = 3 * PLANE2 + (wn * 64 + i32) * LDP2 + kh;
    auto read_frags = [&](Frags& f, const u16* s) {
#pragma unroll
        for (int p = 2; p >= 0; --p) {                        // l planes first: the first products need them
#pragma unroll
            for (int a = 0; a < 2; ++a) f.a[p][a] = *reinterpret_cast<const bf16x8*>(&s[p * PLANE2 + fa_o + a * 32 * LDP2]);
#pragma unroll
            for (int b = 0; b < 2; ++b) f.b[p][b] = *reinterpret_cast<const bf16x8*>(&s[p * PLANE2 + fb_o + b * 32 * LDP2]);
        }
    };
    constexpr int PA[8] = {2, 1, 0, 2, 1, 0, 1, 0};           // a piece / b piece per product, smallest terms first;
    constexpr int PB[8] = {1, 2, 2, 0, 1, 1, 0, 0};           // the last two (mh, hh) need only a.m, a.h, b.h
    constexpr int FIRST = NPROD == 8 ? 0 : (NPROD == 6 ? 2 : 5);
    auto products = [&](const Frags& f, auto Q0, auto Q1) {
#pragma unroll
        for (int q = decltype(Q0)::value; q < decltype(Q1)::value; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if constexpr (ABL & 16) asm volatile("" :: "v"(f.a[PA[q]][a]), "v"(f.b[PB[q]][b]));
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[q]][a], f.b[PB[q]][b], acc[a][b], 0, 0, 0);
                }
    };
    using std::integral_constant;
    constexpr int NQ1 = 6 - FIRST;                            // products before the barrier
    // one stage: multiply tile t out of `cur`; tile t+1 (in the staging registers) goes to `nxt`; tile t+2 is loaded
    // register set (t + j) % NSET holds tile t + j; the set of tile t is free and takes tile t + NSET
    auto stage = [&](Frags& f, Frags& fn, u16* cur, u16* nxt, int t, auto R) {
        constexpr int r = decltype(R)::value;
        load_stage(integral_constant<int, r>{}, (t + NSET) * BK2);
        store_stage(integral_constant<int, (r + 1) % NSET>{}, nxt);
        products(f, integral_constant<int, FIRST>{}, integral_constant<int, 6>{});
        if constexpr (PIN) {
#pragma unroll
            for (int i = 0; i < NQ1 * 4; ++i) {
                SGB(M_MFMA, 1);
                SGB(M_VALU, (80 + NQ1 * 4 - 1) / (NQ1 * 4));
                SGB(M_DS_W, 1);
                SGB(M_VMEM_R, 1);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's LDS writes have landed
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        read_frags(fn, nxt);
        products(f, integral_constant<int, 6>{}, integral_constant<int, 8>{});
        if constexpr (PIN) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                SGB(M_MFMA, 1);
                SGB(M_DS_R, 2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nst = K / BK2;                                  // even (K padded to 32)
    u16* s0 = sm;
    u16* s1 = sm + 6 * PLANE2;
    load_stage(integral_constant<int, 0>{}, 0);
    store_stage(integral_constant<int, 0>{}, s0);
    static_assert(NSET == 2 || NSET == 3, "sets");
    load_stage(integral_constant<int, 1>{}, BK2);
    if constexpr (NSET == 3) load_stage(integral_constant<int, 2>{}, 2 * BK2);
    __syncthreads();
    Frags f0, f1;
    read_frags(f0, s0);
    if constexpr (NSET == 2) {
        for (int t = 0; t < nst; t += 2) {
            stage(f0, f1, s0, s1, t, integral_constant<int, 0>{});
            stage(f1, f0, s1, s0, t + 1, integral_constant<int, 1>{});
        }
    } else {
        for (int t = 0; t < nst; t += 6) {                    // nst is a multiple of 6 here (K padded to 96)
            stage(f0, f1, s0, s1, t, integral_constant<int, 0>{});
            stage(f1, f0, s1, s0, t + 1, integral_constant<int, 1>{});
            stage(f0, f1, s0, s1, t + 2, integral_constant<int, 2>{});
            stage(f1, f0, s1, s0, t + 3, integral_constant<int, 0>{});
            stage(f0, f1, s0, s1, t + 4, integral_constant<int, 1>{});
            stage(f1, f0, s1, s0, t + 5, integral_constant<int, 2>{});
        }
    }
    const int hrow = (lane >> 5) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i32;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + a) * 32 + hrow + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * ldc + col] = acc[a][b][r];
            }
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef void (*Kern)(const float*, int, const float*, int, float*, int, int, int, int);
static void run(Kern kern, const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K, int Kp,
                const std::vector<float>& hA, const std::vector<float>& hB) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, dA, Kp, dB, Kp, dC, N, M, N, Kp);
    CK(hipDeviceSynchronize());
    const int it = 50;
    CK(hipEventRecord(e0));
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, dA, Kp, dB, Kp, dC, N, M, N, Kp);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / it;
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    // error against fp64 on sampled rows, in units of 2^-24 * sum |a||b| (the scale of one fp32 rounding of the sum);
    // next to it: the same for a plain fp32 left-to-right FMA sum on the CPU
    double worst = 0, worst32 = 0, rms = 0, rms32 = 0;
    int cnt = 0;
    for (int s = 0; s < 48; ++s) {
        const int r = (int)(((long long)s * 7919 + 13) % M);
        for (int c = 0; c < N; ++c) {
            double ref = 0, mag = 0;
            float f = 0.f;
            for (int k = 0; k < K; ++k) {
                const double a = hA[(size_t)r * Kp + k], b = hB[(size_t)c * Kp + k];
                ref += a * b; mag += std::fabs(a * b);
                f = std::fmaf((float)a, (float)b, f);
            }
            const double u = mag * std::ldexp(1.0, -24);
            const double e = std::fabs(hC[(size_t)r * N + c] - ref) / u, e32 = std::fabs((double)f - ref) / u;
            worst = std::max(worst, e); worst32 = std::max(worst32, e32);
            rms += e * e; rms32 += e32 * e32; ++cnt;
        }
    }
    printf("%-8s %8.1f us  %7.1f TF(fp32-equivalent)   err/(2^-24 sum|ab|): max %.3f rms %.4f   | cpu fp32 fma chain: max %.3f rms %.4f\n",
           name, us, 2.0 * M * N * K / us * 1e-6, worst, std::sqrt(rms / cnt), worst32, std::sqrt(rms32 / cnt));
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 24495, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 831;
    const int Kp = (K + 95) / 96 * 96;
    std::vector<float> hA((size_t)M * Kp, 0.f), hB((size_t)N * Kp, 0.f);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) hA[(size_t)i * Kp + k] = rnd() * (1.0f + (k % 7)) ;
    for (int i = 0; i < N; ++i) for (int k = 0; k < K; ++k) hB[(size_t)i * Kp + k] = rnd() * 0.1f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    printf("M %d N %d K %d (padded %d)\n", M, N, K, Kp);
    run(gemm_split_nt2<6>, "v2 x6", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt3<3, 0>, "v3 x3", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt3<6, 0>, "v3 x6", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt3<6, 1>, "v3 x6p", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<3, 1, 2>, "v4 x3 s2", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2>, "v4 x6 s2", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<3, 1, 3>, "v4 x3 s3", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 3>, "v4 x6 s3", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2, 1>, "  no split", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2, 2>, "  no ds_write", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2, 4>, "  no barrier", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2, 8>, "  no gload", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2, 16>, "  no mfma", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2, 10>, "  no gload/dsw", dA, dB, dC, M, N, K, Kp, hA, hB);
    run(gemm_split_nt4<6, 1, 2, 14>, "  only mfma+dsr", dA, dB, dC, M, N, K, Kp, hA, hB);
    return 0;
}
