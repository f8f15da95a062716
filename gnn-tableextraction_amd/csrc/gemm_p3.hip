// fp32 GEMMs on the bf16 matrix pipe with PRODUCER-SPLIT operands ("planes" GEMMs; the format is csrc/p3.h).
//
// replaces (reference src/components/graphs/models.py): :63 nn.Linear(2F, out) and its autograd (dX, dW) -- the same products
// as csrc/sage_linear.hip / csrc/gemm_split.h, with the operands already cut into three bf16 planes by the kernel that wrote
// them (aggregation / LayerNorm epilogues, the optimiser's tail, the batch assembly).
//
// Why a third GEMM: gemm_split.h cuts every fp32 tile into its bf16 pieces when the tile goes registers -> LDS, i.e. once per
// workgroup that touches it (the weights: ~190 times per launch) -- ~100 VALU instructions per K stage share the issue port
// with 24 MFMAs (profiles/r02/c_pmc_gemm_split.txt: 6.6 VALU per MFMA, matrix pipe busy 28-48 %).  Here a K stage is
// 6 LDS-DMA instructions (buffer_load ... lds: memory -> LDS, no registers, no VALU), 12 fragment reads and 24 MFMAs per wave.
// Same pieces, same six products per fragment pair in the same order (hl, lh, mm, hm, mh, hh), fp32 accumulate: the result is
// bit-identical to gemm_split_kernel on the fp32 operands (tests/test_gemm_p3.py).
//
// Roofline: MFMA bf16 (2.5 PF dense; six products per fp32 product => 417 TF fp32-equivalent).
//
// NT kernel (K = feature index; both operands K-contiguous: C = A B^T, A = [M rows], B = [N rows]):
//   block tile BM x BN (128 x 128 or 64 x 128), 4 waves as 2 x 2, stage = 16 k = one P3 block = 96 bytes per row.
//   LDS image of a stage: [rows][96 bytes], filled by LDS-DMA in linear order (lane -> 16-byte slot); the two 16-byte halves
//   of a plane are swapped in rows with bit 3 set (on the SOURCE address, the destination of an LDS-DMA is linear): the 16
//   lanes a ds_read_b128 is served in then hit 16 distinct 16-byte bank groups (row stride 96 bytes = 24 banks).
// TN kernel (K = row index; dW = dZ^T X, split over the rows): block tile 128 x 128, stage = 16 rows; LDS image
//   [16 k][8 blocks x 96 bytes] in memory order, the block position rotated by 2 (k mod 4) (again on the source side) so the
//   four k rows a transposing read (ds_read_b64_tr_b16) touches fall on disjoint banks.
#include "gte_common.h"
#include "p3.h"
#include "smallk_step.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned SRD_FLAGS = 0x00020000u;      // gfx9 raw buffer, 32-bit data format

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

struct P3Gemm {
    const char* A1; const char* A2; const char* B;       // P3 images (csrc/p3.h)
    const char* An2; const char* Bn2;                    // TN: operands of N segment 1 (nullptr: those of segment 0)
    long long lda1, lda2, ldb, ldan2, ldbn2;             // row strides in bytes
    long long bsa1, bsa2, bsb, bsan2, bsbn2;             // block strides in bytes (96: row-major image; rows * 96: block-major image)
    int KB1, KB2;                                        // NT: 16-wide K blocks of the two A segments (B: KB1 + KB2 blocks per row)
    int M, N, K;                                         // TN: K = rows reduced over
    int Nseg;                                            // TN with two N segments: columns per segment (N == 2 Nseg)
    float* C; long long ldc;
    const float* bias; int bias_cols;                    // bias covers columns [0, bias_cols) (<= 0: all)
    int relu, accumulate;
    int splits, stages_per_split;                        // TN split-K over the rows
    float* slab;                                         // [splits][M][N] (ld = N) when splits > 1
    // row maps (the input layer reads the RESIDENT feature image instead of a per-batch copy of its rows): NT: A1 is the
    // resident image and row m of the product is its row rowsA[m]; TN: B is the resident image and k row k is its row rowsB[k]
    // (entries past K name a row past the image).  res_bytes = size of the image (< 4 GB: 32-bit buffer offsets)
    const int* rowsA; const int* rowsB; long long res_bytes;
    int rows64;                                          // the row map is followed through 64-bit addresses (image >= 4 GB)
    // NT: BOTH K segments are resident images read through rowsA (A2 = the cached aggregated input next to the input itself,
    // gte_gemm_p3_nt_rows2); res_bytes2 = size of the second image
    int rows_both; long long res_bytes2;
    // LayerNorm(+ReLU) backward as the epilogue of an NT product whose tile holds whole rows (N <= 256): the product is
    // dy = d(loss) / d(y of the layer below), never stored; the workgroup writes dz = LN'(z)(mask . dy) as fp32 and as a P3 image
    // and leaves the column partials {sum g xhat, sum g, sum dz} in ln_part[tile][3][N]  (gte_gemm_p3_nt_ln_bwd)
    const float* ln_z; long long ln_ldz; const float* ln_stats; const float* ln_gamma; const float* ln_beta; int ln_relu;
    float* ln_dz; long long ln_lddz; char* ln_dzp3; long long ln_ldp3; float* ln_part;
    // LayerNorm(+ReLU) FORWARD as the epilogue of an NT product whose tile holds whole rows (N <= 256; gte_gemm_p3_nt_ln_fwd): the
    // workgroup writes z = product + bias (fp32, to C: the backward's operand), the row statistics, y = relu?(LN(z)) as fp32
    // (nullable) and as a P3 image (nullable); gamma / beta / relu in the ln_* fields above
    float* lnf_y; long long lnf_ldy; char* lnf_yp3; long long lnf_ldp; float* lnf_stats; float lnf_eps;
    // ... or the WHOLE backward of a short-input layer below (gte_gemm_p3_nt_smallk_bwd): z recomputed from its k1 + k2 <= 28 inputs,
    // LayerNorm backward, dW = dz^T [x | ahn] accumulated per lane; outputs: sk_part_dw[tile][N][K], ln_part[tile][3][N]
    const float* sk_x; long long sk_ldx; int sk_k1; const float* sk_ahn; long long sk_ldahn; int sk_k2;
    const float* sk_W; long long sk_ldw; const float* sk_bias; float* sk_part_dw;
};


// 64 lanes x 16 bytes memory -> LDS (buffer_load_dwordx4 ... lds): lane l's 16 bytes land at dst + 16 l; dst wave-uniform.
// A plain (non-template) function on purpose: inside a kernel TEMPLATE whose body depends on the template parameters the
// HOST pass of hipcc 7.2 failed to instantiate the kernel -- silently: no diagnostic, an undefined symbol at load time.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t srd, char* dst, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)dst, 16, voffset, 0, 0, 0);
}

// the same from a 64-bit per-lane address (global_load_lds_dwordx4): the row-mapped operand of a resident image of 4 GB or more,
// whose rows a 32-bit buffer offset does not reach.  No range check: every lane names valid memory.
typedef const void __attribute__((address_space(1))) * global_cptr_t;
__device__ __forceinline__ void gdma16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((global_cptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
}

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

template <int TM, int TN>
struct Frags { bf16x8 a[3][TM], b[3][TN]; };

template <int TM, int TN>
__device__ __forceinline__ void products(f32x16 (&acc)[TM][TN], const Frags<TM, TN>& f) {
    // the six piece products of a fragment pair, smallest terms first (0 = h, 1 = m, 2 = l): hl, lh, mm, hm, mh, hh
#ifndef P3_ORDER
#define P3_ORDER 0        // measurement builds: 1 = A fragment held over consecutive MFMAs, 2 = B fragment held
#endif
#if P3_ORDER == 0
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0};
    constexpr int PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[q]][a], f.b[PB[q]][b], acc[a][b], 0, 0, 0);
#elif P3_ORDER == 1
    constexpr int PA[6] = {2, 1, 1, 0, 0, 0};
    constexpr int PB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[q]][a], f.b[PB[q]][b], acc[a][b], 0, 0, 0);
#else
    constexpr int PA[6] = {0, 1, 0, 2, 1, 0};
    constexpr int PB[6] = {2, 1, 1, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < TM; ++a)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[q]][a], f.b[PB[q]][b], acc[a][b], 0, 0, 0);
#endif
}

// (measurement builds: the fragments count as used)
template <int TM, int TN>
__device__ __forceinline__ void keep_frags(const Frags<TM, TN>& f) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int a = 0; a < TM; ++a) asm volatile("" ::"v"(f.a[pl][a]));
#pragma unroll
        for (int b = 0; b < TN; ++b) asm volatile("" ::"v"(f.b[pl][b]));
    }
}

// Store epilogue of a wave's TM x TN accumulator tiles through a buffer descriptor over the output: rows past M and columns
// past `cols_valid` get an offset outside the window and are dropped by the range check (branch-free; sage_linear.hip).
// The stored value goes through a VGPR on purpose: with an accumulator element as the store's data operand hipcc 7.2 emitted
// element 0 for all sixteen stores (gemm_bf16.hip).
template <int TM, int TN>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[TM][TN], float* outp, long long ldo, int M, int cols_total,
                                           int row0, int col0, int cols_valid_end, const float* bias, int bias_cols, int relu,
                                           int accumulate, int lane) {
    const int col_l = lane & 31, hrow = (lane >> 5) * 4;
    const __amdgpu_buffer_rsrc_t c_srd =
        __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)(((long long)(M - 1) * ldo + cols_total) * 4), SRD_FLAGS);
    const int ld4 = (int)ldo * 4;
    const bool post = accumulate || relu;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = col0 + b * 32 + col_l;
        const bool cok = col < cols_valid_end;
        const int coff = cok ? col * 4 : (int)0x80000000;
        const float bv = (bias && cok && (bias_cols <= 0 || col < bias_cols)) ? bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int roff0 = (row0 + a * 32 + hrow) * ld4 + coff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = roff0 + ((r & 3) + 8 * (r >> 2)) * ld4;
                float v = acc[a][b][r] + bv;
                if (post) {
                    if (accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_srd, off, 0, 0));
                    if (relu) v = fmaxf(v, 0.f);
                }
                asm volatile("" : "+v"(v));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_srd, off, 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NT: C[m][n] = sum_k A[m][k] B[n][k]
// ---------------------------------------------------------------------------------------------------------------
// s_waitcnt vmcnt(N) lgkmcnt(0) + s_barrier, N a compile-time count of LDS-DMA instructions allowed to stay in flight
template <int N>
__device__ __forceinline__ void wait_dma_barrier() {
    static_assert(N >= 0 && N <= 24, "vmcnt");
#define GTE_W(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    GTE_W(1); GTE_W(2); GTE_W(3); GTE_W(4); GTE_W(5); GTE_W(6); GTE_W(7); GTE_W(8); GTE_W(9); GTE_W(10); GTE_W(11); GTE_W(12);
    GTE_W(13); GTE_W(14); GTE_W(15); GTE_W(16); GTE_W(17); GTE_W(18); GTE_W(19); GTE_W(20); GTE_W(21); GTE_W(22); GTE_W(23); GTE_W(24);
#undef GTE_W
}

#ifndef P3_SCHED
#define P3_SCHED 1        // 0: the compiler's own order inside a stage
#endif
#define GTE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x008, SG_VMEM_R = 0x020, SG_DS_R = 0x100;      // (sched_group_barrier masks: 0x002 VALU, 0x004 SALU, 0x200 DS write)
#ifndef P3_ABL
#define P3_ABL 0          // measurement builds only (profiles/micro/gemm_p3_abl.hip, l0_fwd_abl.hip): 1 no DMA, 2 no barrier, 4 fragments
#endif                    // read once, 8 no MFMA, 16 clock stamps into p.slab, 32 no epilogue, 64 / 128 the A / B requests move nothing
                          // (loader-wave kernel); results are wrong by construction when a bit is set

// WM x WN waves (4 or 8), every wave TM x TN accumulator tiles of 32 x 32; block tile BM x BN = (32 TM WM) x (32 TN WN).
// The stage images live in a ring of NBUF LDS buffers and the LDS-DMA of stage t + NBUF - 1 is requested at the start of
// stage t: with NBUF = 3 two stages are in flight while one is multiplied, and the wait that closes a stage is COUNTED
// (vmcnt = the instructions of the newest NBUF - 2 stages), never a drain.  Every wave issues the same number NI of DMA
// instructions per stage (the image is padded to NI * waves KiB; the padding instructions are out of range of their
// descriptor and move nothing), so the count is a compile-time constant.
// What the tile size buys: a stage moves (BM + BN) * 96 bytes into LDS for BM * BN * 16 * 2 * 6 flop, i.e. at the full matrix
// rate a CU has to take in 2048 (1 / BM + 1 / BN) bytes per cycle: 32 at 128 x 128, 18.7 at 192 x 256, 16 at 256 x 256.
// ---- epilogue of an NT tile that holds whole rows (BN = 256): the whole backward of a short-input layer below ------------------
// (csrc/smallk_step.h; called by the ring kernel with every wave of the workgroup behind the ring's last barrier)
template <int TM, int TN, int WM, int WN, int LDS_BYTES>
__device__ __forceinline__ void smallk_bwd_epilogue(const f32x16 (&acc)[TM][TN], char* lds, const P3Gemm& p, int m0, unsigned lb, int wave,
                                                    int wm, int wn, int lane, int tid) {
    constexpr int NW = WM * WN, BM = WM * TM * 32, KMAX = 28, KC = KMAX / 4, XR = 32;     // XR: rows whose inputs are staged at a time
    static_assert(WN * TN * 32 == 256 && TN == 2 && NW == 8, "whole rows per workgroup, eight waves");
    // the WHOLE tile goes to LDS at once (every wave's accumulators are free before the 4 x 28 dW accumulators come alive:
    // with half the tile at a time the other half's accumulators and the dW sums did not fit 256 registers)
    static_assert((BM * 256 + KMAX * 256 + XR * KMAX) * 4 <= LDS_BYTES, "LDS");
    // (the two stages requested past the end are empty windows, and an out-of-range LDS-DMA lane WRITES a zero: every wave's
    // requests have landed before any wave re-uses the stage images)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    float* tile = reinterpret_cast<float*>(lds);                   // [BM][256]
    float* Wt = tile + BM * 256;                                   // [Kp][256] (of [KMAX][256]), rows K .. Kp-1 zero
    float* xs = Wt + KMAX * 256;                                   // [XR][Kp]
    const int n = p.N, M = p.M, K = p.sk_k1 + p.sk_k2, Kp = (K + 3) & ~3, ns = 256;
    const int j4 = 4 * lane;
    const bool okc = j4 < n;
    const int col_l = lane & 31, hrow = (lane >> 5) * 4;
    {
        float wv[KMAX];                                            // W^T: thread c < 256 walks row c of W
        const float* wr = p.sk_W + (long long)min(tid, n - 1) * p.sk_ldw;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) wv[k] = k < K ? wr[k] : 0.f;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[(wm * TM * 32 + a * 32 + hrow + (r & 3) + 8 * (r >> 2)) * 256 + wn * 64 + b * 32 + col_l] = acc[a][b][r];
        if (tid < 256) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < Kp) Wt[k * ns + tid] = tid < n ? wv[k] : 0.f;
        }
    }
    float b4[4] = {0.f, 0.f, 0.f, 0.f}, g4[4] = {0.f, 0.f, 0.f, 0.f}, be4[4] = {0.f, 0.f, 0.f, 0.f};
    if (okc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { b4[e] = p.sk_bias[j4 + e]; g4[e] = p.ln_gamma[j4 + e]; be4[e] = p.ln_beta[j4 + e]; }
    }
    float dw[4][KMAX], s_dg[4] = {0.f, 0.f, 0.f, 0.f}, s_db[4] = {0.f, 0.f, 0.f, 0.f}, s_dbias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) dw[e][k] = 0.f;
    const float inv_n = 1.0f / (float)n;
    const float* wl = Wt + (okc ? j4 : 0);
    for (int c0 = 0; c0 < BM; c0 += XR) {
        const int row_base = m0 + c0;
        if (row_base >= M) break;                                  // uniform
        float xv[KMAX / 4];                                        // the chunk's inputs: 4 threads per row
        const int rlx = tid >> 2, part = tid & 3;
        if (rlx < XR) {
            const int r = min(row_base + rlx, M - 1);
            const float* r1 = p.sk_x + (long long)r * p.sk_ldx;
            const float* r2 = p.sk_ahn ? p.sk_ahn + (long long)r * p.sk_ldahn : r1;
#pragma unroll
            for (int i = 0; i < KMAX / 4; ++i) {
                const int k = part + 4 * i;
                xv[i] = 0.f;
                if (k < p.sk_k1) xv[i] = r1[k];
                else if (k < K) xv[i] = r2[k - p.sk_k1];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // tile / W^T written; the previous chunk's readers are done
        if (rlx < XR) {
#pragma unroll
            for (int i = 0; i < KMAX / 4; ++i) {
                const int k = part + 4 * i;
                if (k < Kp) xs[rlx * Kp + k] = xv[i];
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int rl0 = 2 * wave; rl0 < XR; rl0 += 2 * NW) {         // two rows per step
            float gy[2][4], mean[2], rstd[2];
            bool rok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rg = row_base + rl0 + u;
                rok[u] = rg < M;
                const int rc = min(rg, M - 1);
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (okc && rok[u]) t = *reinterpret_cast<const float4*>(tile + (c0 + rl0 + u) * 256 + j4);
                gy[u][0] = t.x; gy[u][1] = t.y; gy[u][2] = t.z; gy[u][3] = t.w;
                mean[u] = p.ln_stats[rc]; rstd[u] = p.ln_stats[M + rc];
            }
            if (!rok[0]) break;                                     // wave-uniform
            gte_smallk_bwd_step<KMAX, 2>(gy, mean, rstd, rok, okc, xs + rl0 * Kp, Kp, wl, ns, b4, g4, be4, p.ln_relu, inv_n, dw, s_dg, s_db,
                                         s_dbias);
        }
    }
    // the eight waves' partial sums through LDS, a quarter of the k range per round (csrc/smallk_bwd.hip), into the tile's [N][K]
    // result in LDS; global stores last, coalesced
    float* red = reinterpret_cast<float*>(lds);                    // [NW][KC][256]
    float* obuf = red + NW * KC * 256;                             // [N][K]
    static_assert((NW * KC * 256 + 256 * KMAX) * 4 <= LDS_BYTES, "LDS");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c * KC < Kp) {                                          // uniform
            if (okc) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    const int k = c * KC + kk;
                    float4 v; v.x = dw[0][k]; v.y = dw[1][k]; v.z = dw[2][k]; v.w = dw[3][k];
                    *reinterpret_cast<float4*>(red + (wave * KC + kk) * 256 + j4) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid < n) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    const int k = c * KC + kk;
                    if (k < K) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) v += red[(w * KC + kk) * 256 + tid];
                        obuf[tid * K + k] = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[(wave * 3 + 0) * 256 + j4 + e] = s_dg[e];
        red[(wave * 3 + 1) * 256 + j4 + e] = s_db[e];
        red[(wave * 3 + 2) * 256 + j4 + e] = s_dbias[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float* pp = p.ln_part + (long long)lb * 3 * n;
    for (int i = tid; i < 3 * 256; i += NW * 64) {
        const int q = i >> 8, j = i & 255;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * 3 + q) * 256 + j];
        if (j < n) pp[q * n + j] = v;
    }
    float* pd = p.sk_part_dw + (long long)lb * n * K;
    for (int e = tid; e < n * K; e += NW * 64) pd[e] = obuf[e];
}

template <int WM, int WN, int TM, int TN, int NBUF, int WGS, int EPI = 0>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN * WGS + 3) / 4)
gemm_p3_nt_ring_kernel(const P3Gemm p) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_INST = BM * 96 / 1024, B_INST = BN * 96 / 1024, N_INST = A_INST + B_INST;
    constexpr int A_BYTES = BM * 96;
    constexpr int NI = (N_INST + NW - 1) / NW;
    constexpr int STAGE = NI * NW * 1024;
    static_assert((BM * 96) % 1024 == 0 && (BN * 96) % 1024 == 0, "tile rows");
    extern __shared__ __attribute__((aligned(16))) char lds[];             // NBUF stage images

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    const unsigned lb = gte_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = (int)(lb / (unsigned)tiles_n), tn = (int)(lb % (unsigned)tiles_n);
    const int m0 = tm * BM, n0 = tn * BN;
    const int rowsA = min(BM, p.M - m0), rowsB = min(BN, p.N - n0);

    const int lda1 = (int)p.lda1, lda2 = (int)(p.A2 ? p.lda2 : p.lda1), ldb = (int)p.ldb;
    const char* baseA1 = p.rowsA ? p.A1 : p.A1 + (long long)m0 * p.lda1;
    const char* baseA2 = p.A2 ? (p.rows_both ? p.A2 : p.A2 + (long long)m0 * p.lda2) : baseA1;
    const int recA1 = p.rowsA ? (int)(unsigned)p.res_bytes : rowsA * (int)p.lda1;     // window of segment 1
    const int recA2 = p.rows_both ? (int)(unsigned)p.res_bytes2 : rowsA * lda2;       // ... of segment 2
    const char* baseB = p.B + (long long)n0 * p.ldb;
    const long long bsa1 = p.bsa1, bsa2 = p.A2 ? p.bsa2 : p.bsa1, bsb = p.bsb;

    // LDS-DMA slots: instruction ii of a stage fills bytes [1024 ii, 1024 ii + 1024) of the stage image (A rows, then B rows,
    // then padding); lane -> slot s = 64 ii + lane = (row s / 6, part s % 6), part = 2 plane + half; the source half is swapped
    // in rows with bit 3 set
    constexpr int OOB = 0x7f000000;                  // a vector offset past every window (rows x stride < 2^30)
    int vo1[NI], vo2[NI], isb[NI];
    static_for<NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int ii = i * NW + wave;
        const bool b = ii >= A_INST;
        const int s = (b ? ii - A_INST : ii) * 64 + lane, row = s / 6, part = s - row * 6, sp = part ^ ((row >> 3) & 1);
        isb[i] = b ? 1 : 0;
        vo1[i] = ii >= N_INST ? OOB : row * (b ? ldb : lda1) + sp * 16;
        vo2[i] = ii >= N_INST ? OOB : row * (b ? ldb : lda2) + sp * 16;
        if (p.rowsA && !b && ii < N_INST) {             // row m0 + row of the product = row rowsA[.] of the resident image
            const int rr = row < rowsA ? p.rowsA[m0 + row] : -1;
            vo1[i] = rr < 0 ? (int)0xfffffff0u : (int)((unsigned)rr * (unsigned)lda1 + (unsigned)(sp * 16));
            if (p.rows_both) vo2[i] = rr < 0 ? (int)0xfffffff0u : (int)((unsigned)rr * (unsigned)lda2 + (unsigned)(sp * 16));
        }
    });
    const int KB1 = p.KB1, T = p.KB1 + p.KB2;
    // the window of a stage: base moved to the stage's K block on the scalar unit, rows past the tile's valid rows (and every
    // row of a stage past the end of K) out of range
    auto issue = [&](int t, char* buf) {
        const bool seg = t >= KB1;
        const int live = t < T ? 1 : 0;
        const char* pa = seg ? baseA2 + (long long)(t - KB1) * bsa2 : baseA1 + (long long)t * bsa1;
        const __amdgpu_buffer_rsrc_t sa =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pa), 0, (seg ? recA2 : recA1) * live, SRD_FLAGS);
        const __amdgpu_buffer_rsrc_t sb =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseB + (long long)t * bsb), 0, rowsB * ldb * live, SRD_FLAGS);
        static_for<NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            dma16(isb[i] ? sb : sa, buf + (i * NW + wave) * 1024, seg ? vo2[i] : vo1[i]);
        });
    };

    // fragment of plane pl, 32-row block blk: lane (r = lane & 31, h = lane >> 5) reads the 16 bytes (row r, k = 8 h .. 8 h + 7)
    const int r = lane & 31, hs = ((lane >> 5) ^ ((r >> 3) & 1)) * 16;
    const int a_rd = (wm * TM * 32 + r) * 96 + hs;
    const int b_rd = A_BYTES + (wn * TN * 32 + r) * 96 + hs;
    using F = Frags<TM, TN>;
    auto read_frags = [&](F& f, const char* buf) {
        // in the order the products need them: (a_h, b_l), (a_l, b_h), (a_m, b_m)
        static_for<3>([&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int oa = g == 0 ? 0 : (g == 1 ? 2 : 1), ob = g == 0 ? 2 : (g == 1 ? 0 : 1);
            static_for<TM>([&](auto A) {
                constexpr int a = decltype(A)::value;
                f.a[oa][a] = *reinterpret_cast<const bf16x8*>(buf + a_rd + a * 32 * 96 + oa * 32);
            });
            static_for<TN>([&](auto B) {
                constexpr int b = decltype(B)::value;
                f.b[ob][b] = *reinterpret_cast<const bf16x8*>(buf + b_rd + b * 32 * 96 + ob * 32);
            });
        });
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    int next = 0, wr = 0;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d) {
        issue(next, lds + wr * STAGE);
        ++next;
        wr = wr + 1 == NBUF ? 0 : wr + 1;
    }
    wait_dma_barrier<(NBUF - 2) * NI>();
    int rd = 0;
#if P3_ABL & 16
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
#if P3_ABL & 4
    F f;
    read_frags(f, lds);
#endif
    for (int t = 0; t < T; ++t) {
#if !(P3_ABL & 1)
        issue(next, lds + wr * STAGE);
#endif
        ++next;
        wr = wr + 1 == NBUF ? 0 : wr + 1;
#if !(P3_ABL & 4)
        F f;
        read_frags(f, lds + rd * STAGE);
#endif
#if P3_ABL & 8
        keep_frags<TM, TN>(f);
#else
        products<TM, TN>(acc, f);
#endif
#if P3_SCHED
        // Issue order of a stage, pinned: the fragments of the first two products, then per product its TM x TN MFMAs with
        // one LDS-DMA instruction of the next-but-one stage in front (a burst of NI DMA instructions at the head of the stage
        // held the wave's issue port for several hundred cycles with the matrix pipe idle) and the third product's fragments
        // behind the first product.
        GTE_SGB(SG_DS_R, 2 * (TM + TN));
        static_for<6>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            constexpr int ndma = (NI * (q + 1)) / 6 - (NI * q) / 6;
            if constexpr (ndma > 0 && !(P3_ABL & 1)) GTE_SGB(SG_VMEM_R, ndma);
            GTE_SGB(SG_MFMA, TM * TN);
            if constexpr (q == 0) GTE_SGB(SG_DS_R, TM + TN);
        });
#endif
#if P3_ABL & 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        wait_dma_barrier<(NBUF - 2) * NI>();
#endif
#if P3_SCHED
        __builtin_amdgcn_sched_barrier(0);
#endif
        rd = rd + 1 == NBUF ? 0 : rd + 1;
    }
#if P3_ABL & 16
    if (tid == 0 && p.slab) {
        long long* st = reinterpret_cast<long long*>(p.slab) + 2 * blockIdx.x;
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#endif
    if constexpr (EPI == 2)
        smallk_bwd_epilogue<TM, TN, WM, WN, 160 * 1024>(acc, lds, p, m0, lb, wave, wm, wn, lane, tid);     // (launched with 160 KB of LDS)
    else
        store_tile<TM, TN>(acc, p.C, p.ldc, p.M, p.N, m0 + wm * TM * 32, n0 + wn * TN * 32, p.N, p.bias, p.bias_cols, p.relu,
                           p.accumulate, lane);
}

// (A variant of the ring kernel on v_mfma_f32_16x16x32_bf16 -- two plane products per instruction -- was measured in round 3 and
// removed in round 4: same matrix-pipe time, no clock advantage at these tile shapes; profiles/r03/gemm_p3.md section 5.)

// ---- epilogues of a loader-wave tile (compute waves only; the loader waves have left, or wait at the epilogue's first barrier) ----
// LNB: 0 plain store; 1 / 3 LayerNorm(+ReLU) backward of the tile's rows (3: a width that is not a multiple of 16); 4 LayerNorm(+ReLU)
// forward.  The LayerNorm forms need whole rows per workgroup (BN = 256) and re-use the stage images as the tile's row slice.
template <int WM, int WN, int TM, int TN, int LNB>
__device__ __forceinline__ void lw_epilogue(const f32x16 (&acc)[TM][TN], char* lds, const P3Gemm& p, int m0, int n0, unsigned lb, int wave,
                                            int lane, int tid) {
    constexpr int NW = WM * WN, BN = WN * TN * 32;
    const int wm = wave / WN, wn = wave % WN;
    if constexpr (LNB == 0) {
        store_tile<TM, TN>(acc, p.C, p.ldc, p.M, p.N, m0 + wm * TM * 32, n0 + wn * TN * 32, p.N, p.bias, p.bias_cols, p.relu,
                           p.accumulate, lane);
    } else if constexpr (LNB == 4) {
        // ---- LayerNorm(+ReLU) FORWARD of the tile's rows: z = product + bias goes to LDS row-major per slice of TM * 32 rows, then
        // every wave takes rows of the slice in the layout and with the arithmetic of ln_relu_fwd_gen_kernel<1, .> (lane l = columns
        // 4 l .. 4 l + 3, two-pass statistics over the n true columns): z, stats, y and the image are bit for bit what
        // gte_gemm_p3_nt + gte_ln_relu_fwd_p3 write ----
        asm volatile("s_barrier" ::: "memory");
        static_assert(BN == 256, "whole rows per workgroup");
        constexpr int SR = TM * 32, LDT = 256;
        float* tile = reinterpret_cast<float*>(lds);
        const int n = p.N, M = p.M;
        const int j4 = 4 * lane;
        const int n16 = (n + 15) & ~15;
        const bool okc = j4 < n;
        float gam[4], bet[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gam[e] = j4 + e < n ? p.ln_gamma[j4 + e] : 0.f; bet[e] = j4 + e < n ? p.ln_beta[j4 + e] : 0.f; }
        const float inv_n = 1.0f / (float)n;
        const int col_l = lane & 31, hrow = (lane >> 5) * 4;
        float bv[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = wn * (TN * 32) + b * 32 + col_l;
            bv[b] = (p.bias && col < n) ? p.bias[col] : 0.f;
        }
        for (int sl = 0; sl < WM; ++sl) {
            asm volatile("s_barrier" ::: "memory");                    // the previous slice's readers are done
            if (wm == sl) {
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            tile[(a * 32 + hrow + (r & 3) + 8 * (r >> 2)) * LDT + wn * (TN * 32) + b * 32 + col_l] = acc[a][b][r] + bv[b];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int row_base = m0 + sl * SR;
            // Two rows per round.  The stores are what this phase waits for (every workgroup of the launch is here at once: ~7 bytes
            // per cycle and CU, issue-bound), so they are made wide: lanes 2 i / 2 i + 1 swap halves and store EIGHT consecutive image
            // columns of ONE row each (16 bytes per plane instead of 8), and the row statistics collect in lanes (lane k = the
            // wave's k-th row) and leave in two stores per wave instead of two per row.  Values and bytes unchanged.
            float st_mean = 0.f, st_rstd = 0.f;
            int kk = 0;
            for (int rl = wave; rl < SR; rl += 2 * NW, kk += 2) {
                if (row_base + rl >= M) break;                          // wave-uniform
                float o[2][4];
                bool rv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int rlu = rl + u * NW;
                    const long long rg = row_base + rlu;
                    rv[u] = rlu < SR && rg < M;                         // wave-uniform
                    float c[4] = {0.f, 0.f, 0.f, 0.f};
                    if (okc && rv[u]) {
                        const float4 t = *reinterpret_cast<const float4*>(tile + rlu * LDT + j4);
                        c[0] = t.x; c[1] = j4 + 1 < n ? t.y : 0.f; c[2] = j4 + 2 < n ? t.z : 0.f; c[3] = j4 + 3 < n ? t.w : 0.f;
                    }
                    float s = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) s += c[e];
                    const float wsum = gte_group_sum<64>(s);           // (z - mean as one fused multiply-add of the row sum, as
                    const float mean = wsum * inv_n;                   // ln_relu_fwd_gen_kernel writes it)
                    float q = 0.f, dv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dv[e] = j4 + e < n ? fmaf(-inv_n, wsum, c[e]) : 0.f; q = fmaf(dv[e], dv[e], q); }
                    const float rstd = rsqrtf(gte_group_sum<64>(q) * inv_n + p.lnf_eps);
                    if (lane == kk + u) { st_mean = mean; st_rstd = rstd; }
                    if (okc && rv[u]) {                                 // z: the operand of the layer's LayerNorm backward
                        f4u t; t.x = c[0]; t.y = c[1]; t.z = c[2]; t.w = c[3];
                        *reinterpret_cast<f4u*>(p.C + rg * p.ldc + j4) = t;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[u][e] = fmaf(fmaf(-inv_n, wsum, c[e]) * rstd, gam[e], bet[e]);
                        if (p.ln_relu) o[u][e] = fmaxf(o[u][e], 0.f);
                        if (j4 + e >= n) o[u][e] = 0.f;
                    }
                    if (p.lnf_y && okc && rv[u]) {
                        f4u t; t.x = o[u][0]; t.y = o[u][1]; t.z = o[u][2]; t.w = o[u][3];
                        *reinterpret_cast<f4u*>(p.lnf_y + rg * p.lnf_ldy + j4) = t;
                    }
                }
                if (p.lnf_yp3) {
                    // even lane: row 0, columns 8 i .. 8 i + 7 = its own four and its neighbour's; odd lane: row 1 likewise
                    const bool odd = lane & 1;
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float got = gte_quad_swap1(odd ? o[0][e] : o[1][e]);
                        x[e] = odd ? got : o[0][e];
                        x[4 + e] = odd ? o[1][e] : got;
                    }
                    const int col0 = 4 * (lane & ~1);
                    if (col0 < n16 && rv[odd ? 1 : 0])
                        p3::store8(p.lnf_yp3 + (long long)(row_base + rl + (odd ? NW : 0)) * p.lnf_ldp, col0, x);
                }
            }
            if (p.lnf_stats) {
                const int rlk = wave + lane * NW;
                const long long rg = row_base + rlk;
                if (lane < kk && rlk < SR && rg < M) { p.lnf_stats[rg] = st_mean; p.lnf_stats[M + rg] = st_rstd; }
            }
        }
    } else {
        // ---- LayerNorm(+ReLU) backward of the tile's rows (the stage images are dead once the loader waves' last requests have
        // landed: they wait for them and join this barrier before they leave) ----
        asm volatile("s_barrier" ::: "memory");
        // Per slice of TM * 32 rows: the waves that own them put their accumulators into LDS row-major, then every wave takes
        // rows of the slice in the layout of ln_relu_bwd_vec_kernel (lane l = columns 4 l .. 4 l + 3, two rows in flight) with
        // ITS arithmetic: dz is bit for bit what the separate launch computes from the stored product.
        static_assert(BN == 256, "whole rows per workgroup");
        constexpr int SR = TM * 32, LDT = 256;                        // slice rows; floats per LDS row
        float* tile = reinterpret_cast<float*>(lds);
        constexpr bool MSK = LNB == 3;
        const int n = p.N, M = p.M;
        const int j4 = 4 * lane;
        // MSK: n is any width; the lane takes part while it has a valid column (the rows of z / dz are padded to a multiple of 4
        // floats), and writes image columns up to the next multiple of 16
        const bool okc = MSK ? j4 < ((n + 3) & ~3) : j4 < n;
        bool oke[4];
        float gam[4], bet[4], s_dg[4], s_db[4], s_dbias[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            oke[e] = MSK ? j4 + e < n : okc;
            gam[e] = oke[e] ? p.ln_gamma[j4 + e] : 1.f;
            bet[e] = oke[e] ? p.ln_beta[j4 + e] : 0.f;
            s_dg[e] = s_db[e] = s_dbias[e] = 0.f;
        }
        const float inv_n = 1.0f / (float)n;
        const int col_l = lane & 31, hrow = (lane >> 5) * 4;
        constexpr int RF = 4;                                           // rows a wave works on at a time
        // One row slice per tile (WM == 1: the 1 x 8 wave layouts): the z rows and statistics of ALL the wave's rows are requested
        // here, before the accumulators go to LDS -- 4 TM rows in flight per wave under the tile write and its two barriers, instead
        // of TM rounds of four rows with a memory latency each (one workgroup per CU: the memory-level parallelism of this phase is
        // all there is)
        // (at 128 rows all 16 x 6 registers beside the 64 accumulators spill: a ring of 8 rows -- the first two rounds' rows are
        // requested here, a round's entries are re-requested for the round after next as soon as it has consumed them)
        constexpr bool PRE = WM == 1;
        constexpr int RPW = PRE ? SR / NW : 1;                          // rows of the tile per wave
        constexpr int PD = RPW <= 12 ? RPW : 8;                         // rows requested ahead
        float zpre[PD][4], mpre[PD], rpre[PD];
        auto request_row = [&](int slot, int k) {                       // row k of this wave -> ring entry `slot`
            const int rg = m0 + wave + k * NW;
            const int rc = rg < M ? rg : min(m0, M - 1);
            f4u zt; zt.x = zt.y = zt.z = zt.w = 0.f;
            if (okc) zt = *reinterpret_cast<const f4u*>(p.ln_z + (long long)rc * p.ln_ldz + j4);
            zpre[slot][0] = zt.x; zpre[slot][1] = zt.y; zpre[slot][2] = zt.z; zpre[slot][3] = zt.w;
            mpre[slot] = p.ln_stats[rc]; rpre[slot] = p.ln_stats[M + rc];
        };
        if constexpr (PRE) {
#pragma unroll
            for (int k = 0; k < PD; ++k) request_row(k, k);
        }
        for (int sl = 0; sl < WM; ++sl) {
            asm volatile("s_barrier" ::: "memory");                    // the previous slice's readers are done
            if (wm == sl) {
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            tile[(a * 32 + hrow + (r & 3) + 8 * (r >> 2)) * LDT + wn * (TN * 32) + b * 32 + col_l] = acc[a][b][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int row_base = m0 + sl * SR;
            constexpr int ROUNDS = (SR + RF * NW - 1) / (RF * NW);
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                const int rl = wave + rd * RF * NW;
                if (rl >= SR) break;
                float gy[RF][4], zz[RF][4], mean[RF], rstd[RF];
                bool rok[RF];
#pragma unroll
                for (int u = 0; u < RF; ++u) {
                    const int rloc = rl + u * NW, rg = row_base + rloc;
                    rok[u] = rloc < SR && rg < M;
                    const int rc = rok[u] ? rg : min(row_base, M - 1);
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bz = a;
                    if (okc) {
                        if (rok[u]) a = *reinterpret_cast<const float4*>(tile + rloc * LDT + j4);
                        if constexpr (!PRE) {
                            const f4u zt = *reinterpret_cast<const f4u*>(p.ln_z + (long long)rc * p.ln_ldz + j4);
                            bz = make_float4(zt.x, zt.y, zt.z, zt.w);
                        }
                    }
                    gy[u][0] = a.x; gy[u][1] = a.y; gy[u][2] = a.z; gy[u][3] = a.w;
                    if constexpr (PRE) {
                        const int k = rd * RF + u;                      // (constant after unrolling: k < RPW because RF divides RPW)
                        const int e_ = (k < RPW ? k : 0) % PD;
#pragma unroll
                        for (int e = 0; e < 4; ++e) zz[u][e] = zpre[e_][e];
                        mean[u] = mpre[e_]; rstd[u] = rpre[e_];
                        if (k + PD < RPW) request_row(e_, k + PD);      // (the entry is free: its row for the round after next)
                    } else {
                        zz[u][0] = bz.x; zz[u][1] = bz.y; zz[u][2] = bz.z; zz[u][3] = bz.w;
                        mean[u] = p.ln_stats[rc]; rstd[u] = p.ln_stats[M + rc];
                    }
                }
                float dd[RF][4];
#pragma unroll
                for (int u = 0; u < RF; ++u) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dd[u][e] = 0.f;
                    if (!rok[u]) continue;                              // wave-uniform
                    const long long rg = row_base + rl + u * NW;
                    float xh[4], g[4];
                    float a = 0.f, b = 0.f;
                    if constexpr (MSK) gte_ln_bwd_pre4m(gy[u], zz[u], mean[u], rstd[u], gam, bet, oke, p.ln_relu, xh, g, a, b);
                    else gte_ln_bwd_pre4(gy[u], zz[u], mean[u], rstd[u], gam, bet, okc, p.ln_relu, xh, g, a, b);
                    const float c1 = gte_group_sum<64>(a) * inv_n, c2 = gte_group_sum<64>(b) * inv_n;
                    if constexpr (MSK) gte_ln_bwd_post4m(g, xh, gam, rstd[u], c1, c2, oke, dd[u], s_dg, s_db, s_dbias);
                    else gte_ln_bwd_post4(g, xh, gam, rstd[u], c1, c2, okc, dd[u], s_dg, s_db, s_dbias);
                    if (okc && p.ln_dz) {                               // (nullable: a consumer that reads the image only)
                        f4u o; o.x = dd[u][0]; o.y = dd[u][1]; o.z = dd[u][2]; o.w = dd[u][3];
                        *reinterpret_cast<f4u*>(p.ln_dz + rg * p.ln_lddz + j4) = o;
                    }
                }
                // the image rows leave two at a time: lanes 2 i / 2 i + 1 swap halves and store EIGHT consecutive columns of one row
                // each (16 bytes per plane; the forward epilogue above has the why)
                if (p.ln_dzp3) {
                    const bool odd = lane & 1;
                    const int col0 = 4 * (lane & ~1);
                    const bool im8 = MSK ? col0 < ((n + 15) & ~15) : col0 < n;
#pragma unroll
                    for (int u = 0; u < RF; u += 2) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float got = gte_quad_swap1(odd ? dd[u][e] : dd[u + 1][e]);
                            x[e] = odd ? got : dd[u][e];
                            x[4 + e] = odd ? dd[u + 1][e] : got;
                        }
                        if (im8 && rok[odd ? u + 1 : u])
                            p3::store8(p.ln_dzp3 + (long long)(row_base + rl + (odd ? u + 1 : u) * NW) * p.ln_ldp3, col0, x);
                    }
                }
            }
        }
        // column partials of the workgroup: the eight waves through LDS, added in wave order
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float* red = reinterpret_cast<float*>(lds);                    // [NW][3][256]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(wave * 3 + 0) * 256 + j4 + e] = s_dg[e];
            red[(wave * 3 + 1) * 256 + j4 + e] = s_db[e];
            red[(wave * 3 + 2) * 256 + j4 + e] = s_dbias[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float* pp = p.ln_part + (long long)lb * 3 * n;
        for (int i = tid; i < 3 * 256; i += NW * 64) {
            const int q = i >> 8, j = i & 255;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(w * 3 + q) * 256 + j];
            if (j < n) pp[q * n + j] = v;
        }
    }
}

// NT with LOADER WAVES: WM x WN compute waves (fragment reads + MFMAs only) and NL loader waves that issue every LDS-DMA
// instruction of the workgroup.  In the ring kernel above each wave spends ~NI x 60-100 issue cycles per stage on its DMA
// instructions, in order, in front of its MFMAs: 21 of 97 us on the layer-0 forward (profiles/r03/gemm_p3.md, ablation "no
// DMA").  A loader wave's stream is: request stage t + 2, wait until stage t + 1 has landed (counted vmcnt), barrier; a compute
// wave's: 3 (TM + TN) fragment reads, 6 TM TN MFMAs, barrier.  One barrier per stage for all waves.
// LNB: 0 plain store; 1 LayerNorm(+ReLU) backward of the tile's rows as the epilogue (3: the same for a LayerNorm width that is
// not a multiple of 16 -- per-element validity, zeros up to the next multiple of 16 in the image); 2 the whole backward of a short-input
// layer below as the epilogue
// BIG: the row-mapped A operand (p.rowsA) is read through 64-bit per-lane addresses -- a resident image of 4 GB or more
template <int WM, int WN, int TM, int TN, int NL, int LNB = 0, bool BIG = false>
__global__ void __launch_bounds__((WM * WN + NL) * 64, (WM * WN + NL + 3) / 4)
gemm_p3_nt_lw_kernel(const P3Gemm p) {
    constexpr int NW = WM * WN, NBUF = 3;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_INST = BM * 96 / 1024, B_INST = BN * 96 / 1024, N_INST = A_INST + B_INST;
    constexpr int A_BYTES = BM * 96;
    constexpr int NI = (N_INST + NL - 1) / NL;                             // DMA instructions per loader wave and stage
    constexpr int STAGE = NI * NL * 1024;
    extern __shared__ __attribute__((aligned(16))) char lds[];             // NBUF stage images

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int tiles_n = (p.N + BN - 1) / BN;
    const unsigned lb = gte_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = (int)(lb / (unsigned)tiles_n), tn = (int)(lb % (unsigned)tiles_n);
    const int m0 = tm * BM, n0 = tn * BN;
    const int T = p.KB1 + p.KB2;

    if (wave >= NW) {
        // ---------------- loader ----------------
        const int lw = wave - NW;
        const int rowsA = min(BM, p.M - m0), rowsB = min(BN, p.N - n0);
        const int lda1 = (int)p.lda1, lda2 = (int)(p.A2 ? p.lda2 : p.lda1), ldb = (int)p.ldb;
        const char* baseA1 = p.rowsA ? p.A1 : p.A1 + (long long)m0 * p.lda1;
        const char* baseA2 = p.A2 ? (p.rows_both ? p.A2 : p.A2 + (long long)m0 * p.lda2) : baseA1;
        const int recA1 = p.rowsA ? (int)(unsigned)p.res_bytes : rowsA * (int)p.lda1;     // window of segment 1
        const int recA2 = p.rows_both ? (int)(unsigned)p.res_bytes2 : rowsA * lda2;       // ... of segment 2
        const char* baseB = p.B + (long long)n0 * p.ldb;
        const long long bsa1 = p.bsa1, bsa2 = p.A2 ? p.bsa2 : p.bsa1, bsb = p.bsb;
        constexpr int OOB = 0x7f000000;
        int vo1[NI], vo2[NI], isb[NI];
        static_for<NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int ii = i * NL + lw;
            const bool b = ii >= A_INST;
            const int s = (b ? ii - A_INST : ii) * 64 + lane, row = s / 6, part = s - row * 6, sp = part ^ ((row >> 3) & 1);
            isb[i] = b ? 1 : 0;
            vo1[i] = ii >= N_INST ? OOB : row * (b ? ldb : lda1) + sp * 16;
            vo2[i] = ii >= N_INST ? OOB : row * (b ? ldb : lda2) + sp * 16;
            if (p.rowsA && !b && ii < N_INST) {             // row m0 + row of the product = row rowsA[.] of the resident image
                const int rr = row < rowsA ? p.rowsA[m0 + row] : -1;
                vo1[i] = rr < 0 ? (int)0xfffffff0u : (int)((unsigned)rr * (unsigned)lda1 + (unsigned)(sp * 16));
                if (p.rows_both) vo2[i] = rr < 0 ? (int)0xfffffff0u : (int)((unsigned)rr * (unsigned)lda2 + (unsigned)(sp * 16));
            }
        });
        const int KB1 = p.KB1;
        // BIG: the lane's piece of its image row as a pointer.  Rows past the tile's valid rows read the tile's first row (their
        // products are never stored), stages past the end of K re-read the last block (never multiplied): no lane leaves the image.
        const char* ap[NI];
        long long ap2d[NI];                                     // rows_both: byte distance from the lane's piece in image 1 to image 2
        if constexpr (BIG)
            static_for<NI>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const int ii = i * NL + lw;
                const int s = ii * 64 + lane, row = s / 6, part = s - row * 6, sp = part ^ ((row >> 3) & 1);
                const int rr = (ii < A_INST) ? p.rowsA[m0 + (row < rowsA ? row : 0)] : 0;
                ap[i] = p.A1 + (long long)rr * p.lda1 + sp * 16;
                ap2d[i] = p.rows_both ? (long long)(reinterpret_cast<uintptr_t>(p.A2) + (unsigned long long)((long long)rr * p.lda2 + sp * 16)) -
                                            (long long)reinterpret_cast<uintptr_t>(ap[i]) : 0;
            });
        auto issue = [&](int t, char* buf) {
            const bool seg = t >= KB1;
            const int live = t < T ? 1 : 0;
            if constexpr (BIG) {
                const __amdgpu_buffer_rsrc_t sbb =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseB + (long long)t * bsb), 0, rowsB * ldb * live, SRD_FLAGS);
                // (stages past the end re-read the last block of the last segment: never multiplied)
                const int tc = t < T ? t : T - 1;
                const bool seg2 = p.rows_both && tc >= KB1;
                const long long ta = seg2 ? (long long)(tc - KB1) * bsa2 : (long long)tc * bsa1;
                static_for<NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if (i * NL + lw < A_INST) gdma16(ap[i] + (seg2 ? ap2d[i] : 0) + ta, buf + (i * NL + lw) * 1024);      // (wave-uniform choice)
                    else dma16(sbb, buf + (i * NL + lw) * 1024, vo1[i]);
                });
                return;
            }
            const char* pa = seg ? baseA2 + (long long)(t - KB1) * bsa2 : baseA1 + (long long)t * bsa1;
            const __amdgpu_buffer_rsrc_t sa =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pa), 0, (seg ? recA2 : recA1) * live, SRD_FLAGS);
            const __amdgpu_buffer_rsrc_t sb =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseB + (long long)t * bsb), 0, rowsB * ldb * live, SRD_FLAGS);
            static_for<NI>([&](auto I) {
                constexpr int i = decltype(I)::value;
#if P3_ABL & (64 | 128)                                      // (measurement: one operand's requests move nothing -- still counted)
                const int vo = ((P3_ABL & 64) && !isb[i]) || ((P3_ABL & 128) && isb[i]) ? OOB : (seg ? vo2[i] : vo1[i]);
                dma16(isb[i] ? sb : sa, buf + (i * NL + lw) * 1024, vo);
#else
                dma16(isb[i] ? sb : sa, buf + (i * NL + lw) * 1024, seg ? vo2[i] : vo1[i]);
#endif
            });
        };
        issue(0, lds);
        issue(1, lds + STAGE);
        wait_dma_barrier<NI>();
        int wr = 2;
        for (int t = 0; t < T; ++t) {
#if !(P3_ABL & 1)
            issue(t + 2, lds + wr * STAGE);
#endif
            wr = wr + 1 == NBUF ? 0 : wr + 1;
            wait_dma_barrier<NI>();
        }
        // an epilogue that re-uses the stage images must not start before the LAST requests have landed: the two stages
        // requested past the end are empty windows, and an out-of-range LDS-DMA lane WRITES a zero
        if constexpr (LNB != 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }
    // ---------------- compute ----------------
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, hs = ((lane >> 5) ^ ((r >> 3) & 1)) * 16;
    const int a_rd = (wm * TM * 32 + r) * 96 + hs;
    const int b_rd = A_BYTES + (wn * TN * 32 + r) * 96 + hs;
    using F = Frags<TM, TN>;
    auto read_frags = [&](F& f, const char* buf) {
        static_for<3>([&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int oa = g == 0 ? 0 : (g == 1 ? 2 : 1), ob = g == 0 ? 2 : (g == 1 ? 0 : 1);
            static_for<TM>([&](auto A) {
                constexpr int a = decltype(A)::value;
                f.a[oa][a] = *reinterpret_cast<const bf16x8*>(buf + a_rd + a * 32 * 96 + oa * 32);
            });
            static_for<TN>([&](auto B) {
                constexpr int b = decltype(B)::value;
                f.b[ob][b] = *reinterpret_cast<const bf16x8*>(buf + b_rd + b * 32 * 96 + ob * 32);
            });
        });
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    asm volatile("s_barrier" ::: "memory");
    int rd = 0;
#if P3_ABL & 16
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
#if P3_ABL & 4
    F f;
    read_frags(f, lds);
#endif
    for (int t = 0; t < T; ++t) {
#if !(P3_ABL & 4)
        F f;
        read_frags(f, lds + rd * STAGE);
#endif
#if P3_ABL & 8
        keep_frags<TM, TN>(f);
#else
        products<TM, TN>(acc, f);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        rd = rd + 1 == NBUF ? 0 : rd + 1;
    }
#if P3_ABL & 16
    if (tid == 0 && p.slab) {
        long long* st = reinterpret_cast<long long*>(p.slab) + 2 * blockIdx.x;
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#endif
#if P3_ABL & 32
    if (p.K != 12345) return;                                  // (measurement: no epilogue)
#endif
    lw_epilogue<WM, WN, TM, TN, LNB>(acc, lds, p, m0, n0, lb, wave, lane, tid);
}

// NT, one column of 256-wide tiles, weights BLOCK-MAJOR (p.ldb == 96: block t of the image is one contiguous [rows][96 bytes]
// run at t * p.bsb): 1 x 8 compute waves of (32 TM) x 32 + NL loader waves.
//   * B never touches LDS: a wave's 32 weight rows of a K block are 3 KB contiguous; lane (r, h) loads its MFMA operand (row r,
//     k = 8 h .. 8 h + 7 of each plane) with three 16-byte buffer loads straight into registers, three K blocks ahead (a ring of four
//     fragment sets).  In the loader-wave kernel above the weights are 2/3 of what a stage moves into LDS and every byte of them
//     is read from LDS exactly once.
//   * A moves through LDS in SLOTS of SQ = 4 K blocks (64 k): [block][row][96 bytes], the stage image of the kernel above four
//     times.  The four 96-byte requests of a row group (1 KB of a block's image) are issued back to back by one loader wave: they
//     touch the same three 128-byte lines of every row, and the vector L1 merges them (profiles/r06/l0_fwd_ablation.txt: 96-byte
//     runs requested a K block at a time fetch every line twice -- 50 GB/s per CU against 82 for whole lines).
//   * ONE barrier per slot (64 k) instead of one per 16 k: inside a slot the two compute waves of a SIMD drift apart and one's
//     fragment reads hide under the other's MFMAs.
// The K blocks of [a1 | a2] form one list cut into slots (a slot may straddle the segments: every request picks its segment's
// window); the LAST slot may hold fewer than SQ blocks: the missing ones are multiplied as zeros (the launcher takes this kernel
// when that pads K by at most 1 / 12).
// Bit-identical to the kernels above: the same six products per fragment pair in the same order, K blocks in the same order.
template <int TM, int NL, int LNB = 0, int SQ = 4>
__global__ void __launch_bounds__((8 + NL) * 64, (8 + NL + 3) / 4)
gemm_p3_nt_sq_kernel(const P3Gemm p) {
    constexpr int WM = 1, WN = 8, TN = 1, NW = 8, NSLOT = 3;
    constexpr int BM = TM * 32, BN = 256;
    constexpr int G = BM * 96 / 1024;                                      // row groups: 1 KB of a block's image each
    constexpr int NG = (G + NL - 1) / NL;                                  // ... per loader wave
    constexpr int NIQ = NG * SQ;                                           // requests per loader wave and slot
    constexpr int BLK = BM * 96, SLOT = SQ * BLK;
    static_assert(SQ == 4 && BLK % 1024 == 0 && NIQ <= 24, "slot");
    extern __shared__ __attribute__((aligned(16))) char lds[];             // NSLOT slots + 1 KB (where padding requests write their zeros)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + BN - 1) / BN;
    const unsigned lb = gte_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = (int)(lb / (unsigned)tiles_n), tn = (int)(lb % (unsigned)tiles_n);
    const int m0 = tm * BM, n0 = tn * BN;
    const int KB1 = p.KB1, T = p.KB1 + p.KB2;
    const int S = (T + SQ - 1) / SQ;                                       // slots over the K blocks of [a1 | a2] as ONE list

    if (wave >= NW) {
        // ---------------- loader ----------------
        const int lw = wave - NW;
        const int rowsA = min(BM, p.M - m0);
        const int lda1 = (int)p.lda1, lda2 = (int)(p.A2 ? p.lda2 : p.lda1);
        const char* baseA1 = p.rowsA ? p.A1 : p.A1 + (long long)m0 * p.lda1;
        const char* baseA2 = p.A2 ? (p.rows_both ? p.A2 : p.A2 + (long long)m0 * p.lda2) : baseA1;
        const int recA1 = p.rowsA ? (int)(unsigned)p.res_bytes : rowsA * lda1;
        const int recA2 = p.rows_both ? (int)(unsigned)p.res_bytes2 : rowsA * lda2;
        constexpr int OOB = (int)0xfffffc00u;                              // past every window, also with a block offset on top
        char* const scratch = lds + NSLOT * SLOT;
        int vo1[NG], vo2[NG];
        static_for<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            const int gi = g * NL + lw;
            const int sl = gi * 64 + lane, row = sl / 6, part = sl - row * 6, sp = part ^ ((row >> 3) & 1);
            vo1[g] = gi >= G ? OOB : row * lda1 + sp * 16;
            vo2[g] = gi >= G ? OOB : row * lda2 + sp * 16;
            if (p.rowsA && gi < G) {
                const int rr = row < rowsA ? p.rowsA[m0 + row] : -1;
                vo1[g] = rr < 0 ? OOB : (int)((unsigned)rr * (unsigned)lda1 + (unsigned)(sp * 16));
                if (p.rows_both) vo2[g] = rr < 0 ? OOB : (int)((unsigned)rr * (unsigned)lda2 + (unsigned)(sp * 16));
            }
        });
        // a slot's K blocks t = SQ s .. SQ s + SQ - 1 of the list; block t lives in segment (t >= KB1) at block t - KB1 (or t): a
        // slot may straddle the two segments -- each request picks its segment's window -- and only the LAST slot can be short
        const __amdgpu_buffer_rsrc_t sa1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseA1), 0, recA1, SRD_FLAGS);
        const __amdgpu_buffer_rsrc_t sa2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseA2), 0, recA2, SRD_FLAGS);
        auto issue = [&](int s, char* slot) {
            static_for<NG>([&](auto GI) {
                constexpr int g = decltype(GI)::value;
                const int gi = g * NL + lw;
                static_for<SQ>([&](auto QI) {
                    constexpr int q = decltype(QI)::value;
#if P3_ABL & 1
                    if (s >= 2) return;
#endif
                    const int t = s * SQ + q;
                    const bool seg = t >= KB1;
                    const int vo = seg ? vo2[g] : vo1[g];
                    dma16(seg ? sa2 : sa1, gi < G ? slot + q * BLK + gi * 1024 : scratch,
                          (t < T && vo != OOB) ? vo + (seg ? t - KB1 : t) * 96 : OOB);
                });
            });
        };
        issue(0, lds);
        issue(1, lds + SLOT);
        wait_dma_barrier<NIQ>();
        int wr = 2;
        for (int s = 0; s < S; ++s) {
            issue(s + 2, lds + wr * SLOT);
            wr = wr + 1 == NSLOT ? 0 : wr + 1;
            wait_dma_barrier<NIQ>();
        }
        if constexpr (LNB != 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }
    // ---------------- compute ----------------
    const int wn = wave;
    const int r = lane & 31, hsel = lane >> 5;
    const int a_rd = r * 96 + (hsel ^ ((r >> 3) & 1)) * 16;
    const int rowsB = min(BN, p.N - n0);
    const char* const baseB = p.B + (long long)n0 * 96;
    const int b_vo = (wn * 32 + r) * 96 + hsel * 16;
    const long long bsb = p.bsb;
    bf16x8 bq[4][3];                                                       // fragment sets of four K blocks in flight
    // (a window per K block: the range check covers scalar offset + lane offset, so the block cannot ride in the scalar offset;
    // invalid = a K block the segment does not have: an empty window, the fragments are zeros)
    auto load_b = [&](bf16x8 (&dst)[3], int t, bool valid) {
        const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(baseB + (long long)min(t, T - 1) * bsb), 0,
                                                                            valid ? rowsB * 96 : 0, SRD_FLAGS);
        static_for<3>([&](auto PL) {
            constexpr int pl = decltype(PL)::value;
            dst[pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(sb, b_vo + pl * 32, 0, 0));
        });
    };
    f32x16 acc[TM][1];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][0][e] = 0.f;
    auto block = [&](const char* blk, const bf16x8 (&b)[3]) {
        bf16x8 fa[3][TM];
        static_for<3>([&](auto GQ) {                                        // planes in the order the products need them: h, l, m
            constexpr int g = decltype(GQ)::value;
            constexpr int oa = g == 0 ? 0 : (g == 1 ? 2 : 1);
            static_for<TM>([&](auto A) {
                constexpr int a = decltype(A)::value;
                fa[oa][a] = *reinterpret_cast<const bf16x8*>(blk + a_rd + a * 32 * 96 + oa * 32);
            });
        });
#if P3_ABL & 8
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int a = 0; a < TM; ++a) asm volatile("" ::"v"(fa[pl][a]));
            asm volatile("" ::"v"(b[pl]));
        }
#else
        // the six piece products, smallest terms first (products<>()): hl, lh, mm, hm, mh, hh
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
        static_for<6>([&](auto QQ) {
            constexpr int q = decltype(QQ)::value;
            static_for<TM>([&](auto A) {
                constexpr int a = decltype(A)::value;
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][a], b[PB[q]], acc[a][0], 0, 0, 0);
            });
        });
#endif
        __builtin_amdgcn_sched_barrier(0);                                  // (no fragment reads hoisted over a whole block: registers)
    };
    // slot s holds K blocks SQ s .. of the list (= of the weight image); the blocks the LAST slot lacks are multiplied as zeros (the
    // loader's requests for them leave the window: zeros in LDS; the weight fragments: load_b): ONE slot body without branches -- with
    // a second, branching copy for a short slot the compiler moved all 48 accumulators through copies at the join and spilled the ring
    auto kfirst = [&](int s) { return s * SQ; };
    auto nvalid = [&](int s) { return s >= S ? 0 : min(SQ, T - s * SQ); };
    int k0 = 0, nv = nvalid(0);
    load_b(bq[0], 0, 0 < nv);
    load_b(bq[1], 1, 1 < nv);
    load_b(bq[2], 2, 2 < nv);
    asm volatile("s_barrier" ::: "memory");
#if P3_ABL & 16
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    int rd = 0;
    for (int s = 0; s < S; ++s) {
        const char* slot = lds + rd * SLOT;
        const int k1 = kfirst(s + 1), nv1 = nvalid(s + 1);
        static_for<SQ>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j + 3 < SQ) load_b(bq[(j + 3) & 3], k0 + j + 3, j + 3 < nv);
            else load_b(bq[(j + 3) & 3], k1 + j + 3 - SQ, j + 3 - SQ < nv1);
            block(slot + j * BLK, bq[j]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        rd = rd + 1 == NSLOT ? 0 : rd + 1;
        k0 = k1; nv = nv1;
    }
#if P3_ABL & 16
    if (tid == 0 && p.slab) {
        long long* st = reinterpret_cast<long long*>(p.slab) + 2 * blockIdx.x;
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#endif
#if P3_ABL & 32
    if (p.K != 12345) return;                                  // (measurement: no epilogue)
#endif
    lw_epilogue<WM, WN, TM, TN, LNB>(acc, lds, p, m0, n0, lb, wave, lane, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// TN: C[m][n] = sum_k A[k][m] B[k][n], split over k
// ---------------------------------------------------------------------------------------------------------------
// WM x WN waves of 64 x 64 (TM = TN = 2): block tile (64 WM) x (64 WN).  4 x 2 waves (256 x 128, one workgroup per CU) take in
// 23 bytes per cycle and CU at the full matrix rate against 32 for 2 x 2 (128 x 128, two workgroups per CU) at the same slab
// bytes per CU: the operand tile of the 256-wide side is shared by eight waves.
#ifndef P3_TN_ABL
#define P3_TN_ABL 0       // measurement builds (profiles/debug/build_variant.sh): 1 no LDS-DMA in the loop, 2 no MFMA, 4 no fragment reads
#endif
// BIG (with MAP): the resident image is 4 GB or more -- its rows are read through 64-bit per-lane addresses, without a range
// check: the map's entries past K name row n_res_rows, which must EXIST and be zero
template <int WM, int WN, int WGS, bool MAP = false, bool BIG = false>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN * WGS + 3) / 4)
gemm_p3_tn_kernel(const P3Gemm p) {
    static_assert(!BIG || MAP, "64-bit rows: the row-mapped kernel");
    static_assert(!MAP || (WM == 2 && WN == 2), "row map: the 128 x 128 tile");
    constexpr int NW = WM * WN, TM = 2, TN = 2;
    constexpr int BM = WM * 64, BN = WN * 64, NBA = BM / 16, NBB = BN / 16;        // 16-feature blocks per k row
    constexpr int RSA = NBA * 96, RSB = NBB * 96;                                  // k-row strides of the stage images
    constexpr int A_BYTES = 16 * RSA, B_BYTES = 16 * RSB;
    constexpr int A_INST = A_BYTES / 1024, B_INST = B_BYTES / 1024, N_INST = A_INST + B_INST;
    constexpr int NI = (N_INST + NW - 1) / NW;                                     // LDS-DMA instructions per wave and stage
    constexpr int STAGE = NI * NW * 1024;
    static_assert((NBA & (NBA - 1)) == 0 && (NBB & (NBB - 1)) == 0 && NBA >= 8 && NBB >= 8, "blocks per row: a power of two >= 8");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nsegs = p.Nseg > 0 ? 2 : 1, seg_cols = p.Nseg > 0 ? p.Nseg : p.N;
    const int seg_tiles = (seg_cols + BN - 1) / BN, tiles_n = nsegs * seg_tiles, tiles_m = (p.M + BM - 1) / BM;
    const unsigned ntile = (unsigned)(tiles_m * tiles_n);
    const unsigned lu = gte_xcd_remap(blockIdx.x, gridDim.x);
    const int split = (int)(lu / ntile);
    const unsigned lb = lu % ntile;
    const int tm = (int)(lb / (unsigned)tiles_n), tn = (int)(lb % (unsigned)tiles_n);
    const int nseg = tn / seg_tiles;
    const int m0 = tm * BM, n0 = (tn % seg_tiles) * BN;
    const char* Am = (nseg && p.An2) ? p.An2 : p.A1;
    const char* Bm = (nseg && p.Bn2) ? p.Bn2 : p.B;
    const int lda = (int)((nseg && p.An2) ? p.ldan2 : p.lda1), ldb = (int)((nseg && p.Bn2) ? p.ldbn2 : p.ldb);

    const int total_stages = (p.K + 15) / 16;
    const int st_begin = split * p.stages_per_split;
    const int st_end = min(total_stages, st_begin + p.stages_per_split);
    const int coa = (m0 >> 4) * 96, cob = (n0 >> 4) * 96;

    // instruction ii of a stage fills bytes [1024 ii, + 1024) of the stage image (A rows, B rows, padding).  Slot s of an
    // operand's image: k row s / (6 NB), position block (s % (6 NB)) / 6, part s % 6; position block jb of row k holds the tile's
    // block (jb - 2 (k & 3)) mod NB (the four k rows of a transposing read then fall on disjoint banks)
    constexpr int OOB = 0x7f000000;
    int vo[NI], isb[NI], kb[NI], vc[NI];
    static_for<NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int ii = i * NW + wave;
        const bool b = ii >= A_INST;
        const int nb = b ? NBB : NBA;
        const int s = (b ? ii - A_INST : ii) * 64 + lane, k = s / (6 * nb), w = s - k * (6 * nb), jb = w / 6, part = w - jb * 6;
        const int fb = (jb - 2 * (k & 3)) & (nb - 1);
        isb[i] = (b && ii < N_INST) ? 1 : 0;            // (padding instructions: any window, offset out of range)
        vo[i] = ii >= N_INST ? OOB : k * (b ? ldb : lda) + fb * 96 + part * 16;
        // B through a row map: the lane's slot lies in k row kb of the stage; it reads that row's resident id itself
        kb[i] = k;
        vc[i] = fb * 96 + part * 16;
    });
    const int* rmap = p.rowsB;
    int rid[3] = {0, 0, 0};                                    // resident rows of the next stage to be issued (this lane's slots)
    if constexpr (MAP)
        if (st_begin < st_end) {
            rid[0] = rmap[st_begin * 16 + kb[3]];
            rid[1] = rmap[st_begin * 16 + kb[4]];
            rid[2] = rmap[st_begin * 16 + kb[5]];
        }
    auto issue = [&](int st, char* buf) {
        const int k0 = st * 16;
        const int rows = st < st_end ? min(16, p.K - k0) : 0;
        const __amdgpu_buffer_rsrc_t sa =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Am + (long long)k0 * lda + coa), 0, max(rows * lda - coa, 0), SRD_FLAGS);
        if constexpr (MAP) {                                   // (2 x 2 waves: instructions 0..2 of a wave are A, 3..5 are B)
            {
                static_assert(!MAP || (A_INST == 12 && NI == 6), "row map: 128 x 128 tile");
                const int live = rows > 0 ? 1 : 0;
                const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(Bm + cob), 0, (int)(unsigned)max(p.res_bytes - cob, 0ll) * live, SRD_FLAGS);
                // the resident row ids of THIS stage were loaded while the previous stage was issued; the loads of the NEXT
                // stage's ids go out first, so that they are older than this stage's LDS-DMA: when they are needed, the counted
                // wait that lets stage st + 1 be issued has long covered them (a load placed after the DMA, or waited for at
                // once -- a scalar load as much as a vector one --, stalls the issue by a memory latency per stage)
                // The ids of THIS stage were requested by the previous issue(), ahead of its six LDS-DMA requests: vmcnt(6) covers
                // them (in the loop the ring's counted wait has passed long ago).  The loads are inline assembly, with their wait
                // here: left to the compiler, the loop-carried ids are copied at the END of a round -- behind a wait for the load
                // just issued and everything older --, volatile loads become system-scope flat loads with vmcnt(0) each, and
                // scalar loads stall the issue by their latency (all three measured / read off the ISA, profiles/r03/gemm_p3.md).
                asm volatile("s_waitcnt vmcnt(6)" : "+v"(rid[0]), "+v"(rid[1]), "+v"(rid[2])::"memory");
                if constexpr (BIG) {
                    const char* g0 = Bm + cob + (long long)rid[0] * ldb + vc[3];
                    const char* g1 = Bm + cob + (long long)rid[1] * ldb + vc[4];
                    const char* g2 = Bm + cob + (long long)rid[2] * ldb + vc[5];
                    const int kn_ = st + 1 < st_end ? k0 + 16 : 0;
                    const int* r0 = rmap + kn_ + kb[3];
                    const int* r1 = rmap + kn_ + kb[4];
                    const int* r2 = rmap + kn_ + kb[5];
                    asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                                 : "=&v"(rid[0]), "=&v"(rid[1]), "=&v"(rid[2])
                                 : "v"(r0), "v"(r1), "v"(r2), "v"(g0), "v"(g1), "v"(g2)       // (the old ids are consumed first)
                                 : "memory");
                    dma16(sa, buf + (0 * NW + wave) * 1024, vo[0]);
                    dma16(sa, buf + (1 * NW + wave) * 1024, vo[1]);
                    dma16(sa, buf + (2 * NW + wave) * 1024, vo[2]);
                    gdma16(g0, buf + (3 * NW + wave) * 1024);
                    gdma16(g1, buf + (4 * NW + wave) * 1024);
                    gdma16(g2, buf + (5 * NW + wave) * 1024);
                    return;
                }
                const int o0 = (int)((unsigned)rid[0] * (unsigned)ldb + (unsigned)vc[3]);
                const int o1 = (int)((unsigned)rid[1] * (unsigned)ldb + (unsigned)vc[4]);
                const int o2 = (int)((unsigned)rid[2] * (unsigned)ldb + (unsigned)vc[5]);
                const int kn = st + 1 < st_end ? k0 + 16 : 0;
                const int* q0 = rmap + kn + kb[3];
                const int* q1 = rmap + kn + kb[4];
                const int* q2 = rmap + kn + kb[5];
                asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                             : "=&v"(rid[0]), "=&v"(rid[1]), "=&v"(rid[2])
                             : "v"(q0), "v"(q1), "v"(q2), "v"(o0), "v"(o1), "v"(o2)       // (o*: the old ids are consumed first)
                             : "memory");
                dma16(sa, buf + (0 * NW + wave) * 1024, vo[0]);
                dma16(sa, buf + (1 * NW + wave) * 1024, vo[1]);
                dma16(sa, buf + (2 * NW + wave) * 1024, vo[2]);
                dma16(sb, buf + (3 * NW + wave) * 1024, o0);
                dma16(sb, buf + (4 * NW + wave) * 1024, o1);
                dma16(sb, buf + (5 * NW + wave) * 1024, o2);
                return;
            }
        }
        const __amdgpu_buffer_rsrc_t sb =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bm + (long long)k0 * ldb + cob), 0, max(rows * ldb - cob, 0), SRD_FLAGS);
        static_for<NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            dma16(isb[i] ? sb : sa, buf + (i * NW + wave) * 1024, vo[i]);
        });
    };

    // transposing fragment reads: lane 32 h + 16 g + 4 q + pp supplies the address of k row 8 h + q (+ 4 for the second read),
    // columns 16 g + 4 pp .. + 3 of the 32-column block; lane i of a 16-lane group receives column i
    const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) a_rd[a] = (8 * h + q) * RSA + ((2 * (wm * TM + a) + g + 2 * q) & (NBA - 1)) * 96 + pp * 8;
#pragma unroll
    for (int b = 0; b < TN; ++b) b_rd[b] = A_BYTES + (8 * h + q) * RSB + ((2 * (wn * TN + b) + g + 2 * q) & (NBB - 1)) * 96 + pp * 8;
    using F = Frags<TM, TN>;
    // The reads are inline assembly: the compiler orders a ds_read_b64_tr_b16 it can see behind ALL LDS-DMA in flight
    // (s_waitcnt vmcnt(0) -- it cannot tell which stage image the read touches), i.e. behind the two stages requested ahead: one
    // memory latency per stage on the critical path.  The ring's own counted wait + barrier is the ordering that is needed.
    typedef char __attribute__((address_space(3))) * lds_char;
    const unsigned lds_base = (unsigned)(__SIZE_TYPE__)(lds_char)lds;
    static_assert(TM == 2 && TN == 2, "the waits below name four fragments per plane pair");
    // fragments of planes (pa of A, pb of B): eight reads
    auto read_pair = [&](F& f, unsigned sbase, auto PA_, auto PB_) {
        static_for<TM + TN>([&](auto X) {
            constexpr int x = decltype(X)::value;
            constexpr bool isb = x >= TM;
            constexpr int pc = isb ? decltype(PB_)::value : decltype(PA_)::value;
            constexpr int lo_off = pc * 32, hi_off = pc * 32 + 4 * (isb ? RSB : RSA);
            const unsigned at = sbase + (unsigned)(isb ? b_rd[isb ? x - TM : 0] : a_rd[isb ? 0 : x]);
            s16x4 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(at), "n"(lo_off) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(at), "n"(hi_off) : "memory");
            const bf16x8 v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            if constexpr (isb) f.b[pc][isb ? x - TM : 0] = v; else f.a[pc][isb ? 0 : x] = v;
        });
    };
    // s_waitcnt lgkmcnt(N) that the fragments of planes (pa, pb) pass through: their consumers stay behind it
    auto wait_pair = [&](F& f, auto N_, auto PA_, auto PB_) {
        constexpr int pa = decltype(PA_)::value, pb = decltype(PB_)::value;
        if constexpr (decltype(N_)::value == 8)
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f.a[pa][0]), "+v"(f.a[pa][1]), "+v"(f.b[pb][0]), "+v"(f.b[pb][1])::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[pa][0]), "+v"(f.a[pa][1]), "+v"(f.b[pb][0]), "+v"(f.b[pb][1])::"memory");
    };
    f32x16 acc[TM][TN];
    auto mfma_pair = [&](const F& f, auto PA_, auto PB_) {
        constexpr int pa = decltype(PA_)::value, pb = decltype(PB_)::value;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[pa][a], f.b[pb][b], acc[a][b], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I8 = std::integral_constant<int, 8>;
    // One stage: the six piece products in the order of products<>() (hl, lh, mm, hm, mh, hh), the 24 fragment reads interleaved
    // so that a product starts as soon as ITS planes have arrived (8 of 24 reads) instead of after all of them: within a wave
    // the LDS latency of a stage hides behind its own first MFMAs (the other wave of the SIMD covers the rest).
    // lgkmcnt is a 4-bit counter: at most 16 reads in flight.
    auto stage_products = [&](int slot) {
        const unsigned sbase = lds_base + (unsigned)slot * STAGE;
        F f;
#if P3_TN_ABL & 4
        static_for<3>([&](auto PC) { f.a[PC.value][0] = f.a[PC.value][1] = f.b[PC.value][0] = f.b[PC.value][1] = bf16x8{}; });
        asm volatile("" : "+v"(f.a[0][0]), "+v"(f.b[0][0]));
#else
        read_pair(f, sbase, I0{}, I2{});
        read_pair(f, sbase, I2{}, I0{});
        wait_pair(f, I8{}, I0{}, I2{});
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(P3_TN_ABL & 2)
        mfma_pair(f, I0{}, I2{});                              // h l
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(P3_TN_ABL & 4)
        read_pair(f, sbase, I1{}, I1{});
        wait_pair(f, I8{}, I2{}, I0{});
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(P3_TN_ABL & 2)
        mfma_pair(f, I2{}, I0{});                              // l h
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(P3_TN_ABL & 4)
        wait_pair(f, I0{}, I1{}, I1{});
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(P3_TN_ABL & 2)
        mfma_pair(f, I1{}, I1{});                              // m m
        mfma_pair(f, I0{}, I1{});                              // h m
        mfma_pair(f, I1{}, I0{});                              // m h
        mfma_pair(f, I0{}, I0{});                              // h h
#else
        asm volatile("" ::"v"(f.a[0][0]), "v"(f.a[1][1]), "v"(f.a[2][0]), "v"(f.b[0][0]), "v"(f.b[1][1]), "v"(f.b[2][1]));
#endif
        __builtin_amdgcn_sched_barrier(0);                     // (the MFMAs are queued BEFORE the wave parks at the ring's barrier)
    };

#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ring of NBUF stage images, two stages requested ahead, counted waits (as the NT kernel)
    constexpr int NBUF = 3;
    int next = st_begin, wr = 0;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d) {
        issue(next, lds + wr * STAGE);
        ++next;
        wr = wr + 1 == NBUF ? 0 : wr + 1;
    }
    wait_dma_barrier<(NBUF - 2) * NI>();
    int rd = 0;
    for (int st = st_begin; st < st_end; ++st) {
#if !(P3_TN_ABL & 1)
        issue(next, lds + wr * STAGE);
#endif
        ++next;
        wr = wr + 1 == NBUF ? 0 : wr + 1;
        stage_products(rd);
        wait_dma_barrier<(NBUF - 2) * NI>();
        rd = rd + 1 == NBUF ? 0 : rd + 1;
    }
    float* outp = p.splits > 1 ? p.slab + (long long)split * p.M * p.N : p.C;
    const long long ldo = p.splits > 1 ? p.N : p.ldc;
    const int cbase = nseg * seg_cols;
#if P3_TN_ABL & 8
    if (p.K == 12345)                                          // (measurement: no slab stores)
#endif
    store_tile<TM, TN>(acc, outp, ldo, p.M, p.N, m0 + wm * TM * 32, cbase + n0 + wn * TN * 32, cbase + seg_cols, nullptr, 0,
                       p.splits > 1 ? 0 : p.relu, p.splits > 1 ? 0 : p.accumulate, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 <-> P3
// ---------------------------------------------------------------------------------------------------------------
// dst (P3, [rows][cols]) element (r, c) = transpose ? src[c * ld + r] : src[r * ld + c].  One thread per half block (8 features).
// Image addressing everywhere below: block fb of row r at r * ldp + fb * bsp (row-major: bsp = 96; block-major: ldp = 96).
__global__ void __launch_bounds__(256)
p3_from_f32_kernel(const float* __restrict__ src, long long ld, int rows, int cols, int transpose, char* __restrict__ dst,
                   long long ldp, long long bsp) {
    const int hb = (int)((cols + 7) / 8);                 // half blocks per row that hold data
    const int hbt = (int)p3::blocks(cols) * 2;            // half blocks per row (the last may be all padding)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * hbt) return;
    const int r = (int)(idx / hbt), j = (int)(idx - (long long)r * hbt);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = j * 8 + e;
        x[e] = 0.f;
        if (j < hb && c < cols) x[e] = transpose ? src[(long long)c * ld + r] : src[(long long)r * ld + c];
    }
    p3::store8(dst + (long long)r * ldp + (long long)(j >> 1) * bsp, (j & 1) * 8, x);
}

// several small matrices in one launch (the weight images of every layer after an optimiser step): blockIdx.y = matrix
constexpr int kMaxP3Batch = 16;
struct P3BatchArgs { gte_p3_desc d[kMaxP3Batch]; };
__global__ void __launch_bounds__(256)
p3_from_f32_batch_kernel(const P3BatchArgs a) {
    const gte_p3_desc d = a.d[blockIdx.y];
    const long long ldp = d.ldp < 0 ? 96 : d.ldp, bsp = d.ldp < 0 ? -d.ldp : 96;
    const int hb = (int)((d.cols + 7) / 8), hbt = (int)p3::blocks(d.cols) * 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < (long long)d.rows * hbt; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / hbt), j = (int)(idx - (long long)r * hbt);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long c = j * 8 + e;
            x[e] = 0.f;
            if (j < hb && c < d.cols) x[e] = d.transpose ? d.src[c * d.ld + r] : d.src[(long long)r * d.ld + c];
        }
        p3::store8(reinterpret_cast<char*>(d.dst) + (long long)r * ldp + (long long)(j >> 1) * bsp, (j & 1) * 8, x);
    }
}

__global__ void __launch_bounds__(256)
p3_to_f32_kernel(const char* __restrict__ src, long long ldp, long long bsp, int rows, int cols, float* __restrict__ dst, long long ld) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * cols) return;
    const int r = (int)(idx / cols), c = (int)(idx - (long long)r * cols);
    dst[(long long)r * ld + c] = p3::load1(src + (long long)r * ldp + (long long)(c >> 4) * bsp, c & 15);
}

// TN tile: 128 x 128 (4 waves, two workgroups per CU).  (The 256 x 128 tile -- 8 waves, one workgroup per CU, a quarter less
// operand traffic per flop -- measured slower on both dW shapes of the step, 126 vs 115 us and 59 vs 55 us, and was removed.)
struct TnPlan { int splits, stages_per_split, splits_bound; };
TnPlan tn_plan(int64_t M, int64_t N, int64_t Nseg, int64_t K) {
    const int cus = gte::device_props().cus;
    const int64_t bm = 128, slots = 2 * cus;
    const int64_t tiles = gte::ceil_div(M, bm) * (Nseg > 0 ? 2 * gte::ceil_div(Nseg, 128) : gte::ceil_div(N, 128));
    const int64_t stages = gte::ceil_div(K > 0 ? K : 1, 16);
    // fill the resident slots exactly or stay below (a straggler round costs a whole unit time); >= 8 stages per split
    int64_t splits = slots / tiles;
    static const int forced = GTE_MEASURE_INT("GTE_P3_TN_SPLITS", 0);
    if (forced > 0) splits = forced;
    static const int min_stages = GTE_MEASURE_INT("GTE_P3_TN_MIN_STAGES", 8);
    if (splits > stages / min_stages) splits = stages / min_stages;
    if (splits < 1) splits = 1;
    TnPlan pl;
    pl.splits_bound = (int)splits;
    pl.stages_per_split = (int)gte::ceil_div(stages, splits);
    pl.splits = (int)gte::ceil_div(stages, pl.stages_per_split);
    return pl;
}

#define GTE_SET_LDS(kernel, bytes) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))

}  // namespace

namespace { int launch_nt(const P3Gemm& p, hipStream_t s); }

extern "C" int64_t gte_p3_row_bytes(int64_t cols) { return cols > 0 ? p3::row_bytes(cols) : 0; }

// (row stride, block stride) of an image from the signed "ldp" of the ABI: a NEGATIVE value names a BLOCK-MAJOR image whose 16-feature
// block fb starts at byte fb * (-ldp) and whose rows follow each other at 96 bytes inside a block (include/gte.h)
namespace {
struct P3Strides { long long ld, bs; bool block_major; };
inline P3Strides p3_strides(int64_t ldp) { return ldp < 0 ? P3Strides{96, -ldp, true} : P3Strides{ldp, 96, false}; }
// is [rows][cols] addressable with these strides?
inline bool p3_strides_ok(int64_t ldp, int64_t rows, int64_t cols) {
    if (ldp < 0) return (-ldp) % 16 == 0 && -ldp >= rows * 96;
    return ldp % 16 == 0 && ldp >= p3::row_bytes(cols);
}
}

extern "C" int gte_p3_from_f32(const float* src, int64_t ld, int64_t rows, int64_t cols, int transpose, void* dst, int64_t ldp,
                               void* stream) {
    if (rows < 0 || cols < 0 || rows > INT32_MAX || cols > INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_from_f32: bad sizes");
    if (rows == 0 || cols == 0) return GTE_OK;
    if (!src || !dst) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_from_f32: null pointer");
    if (ld < (transpose ? rows : cols) || !p3_strides_ok(ldp, rows, cols) || (reinterpret_cast<uintptr_t>(dst) & 15))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_from_f32: leading dimension too small or image not 16-byte aligned");
    const int64_t work = rows * p3::blocks(cols) * 2;
    const P3Strides st = p3_strides(ldp);
    hipLaunchKernelGGL(p3_from_f32_kernel, dim3((unsigned)gte::ceil_div(work, 256)), dim3(256), 0, gte::as_stream(stream), src,
                       (long long)ld, (int)rows, (int)cols, transpose ? 1 : 0, reinterpret_cast<char*>(dst), st.ld, st.bs);
    return gte::check_launch("p3_from_f32");
}

extern "C" int gte_p3_from_f32_batch(const gte_p3_desc* descs, int n, void* stream) {
    if (n < 0 || n > kMaxP3Batch || (n > 0 && !descs)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_from_f32_batch: 0 <= n <= %d", kMaxP3Batch);
    if (n == 0) return GTE_OK;
    P3BatchArgs a;
    int64_t work = 1;
    for (int i = 0; i < n; ++i) {
        const gte_p3_desc& d = descs[i];
        if (!d.src || !d.dst || d.rows <= 0 || d.cols <= 0 || d.ld < (d.transpose ? d.rows : d.cols) || !p3_strides_ok(d.ldp, d.rows, d.cols) ||
            (reinterpret_cast<uintptr_t>(d.dst) & 15))
            return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_from_f32_batch: bad descriptor %d", i);
        a.d[i] = d;
        const int64_t w = d.rows * p3::blocks(d.cols) * 2;
        if (w > work) work = w;
    }
    int64_t bx = gte::ceil_div(work, 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(p3_from_f32_batch_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, gte::as_stream(stream), a);
    return gte::check_launch("p3_from_f32_batch");
}

extern "C" int gte_p3_to_f32(const void* src, int64_t ldp, int64_t rows, int64_t cols, float* dst, int64_t ld, void* stream) {
    if (rows < 0 || cols < 0 || rows > INT32_MAX || cols > INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_to_f32: bad sizes");
    if (rows == 0 || cols == 0) return GTE_OK;
    if (!src || !dst) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_to_f32: null pointer");
    if (ld < cols || !p3_strides_ok(ldp, rows, cols)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "p3_to_f32: leading dimension too small");
    const P3Strides st = p3_strides(ldp);
    hipLaunchKernelGGL(p3_to_f32_kernel, dim3((unsigned)gte::ceil_div(rows * cols, 256)), dim3(256), 0, gte::as_stream(stream),
                       reinterpret_cast<const char*>(src), st.ld, st.bs, (int)rows, (int)cols, dst, (long long)ld);
    return gte::check_launch("p3_to_f32");
}

// Row maps: 32-bit buffer offsets reach images below 4 GB; larger ones are read through 64-bit per-lane addresses
// (gte_gemm_p3_set_rows64(1) forces that path for any size: tests, A/B timing)
// (the three forced configurations of this file -- rows64, the NT tile, the LayerNorm-epilogue row tile -- are THREAD-LOCAL test hooks:
// they affect the calling host thread's launches only; the library keeps no mutable process-wide state besides the GEMM mode)
static thread_local int g_rows64_mode = 0;
static inline int rows64_needed(long long res_bytes) {
    static const int env = GTE_MEASURE_INT("GTE_P3_ROWS64", 0);          // (A/B timing of the train loop)
    return (g_rows64_mode == 1 || env == 1 || res_bytes >= ((long long)1 << 32) - 4096) ? 1 : 0;
}
extern "C" int gte_gemm_p3_set_rows64(int mode) {
    if (mode != 0 && mode != 1) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_set_rows64: mode must be 0 (by size) or 1 (always)");
    g_rows64_mode = mode;
    return GTE_OK;
}

// C[m, n] (+)= [a1 | a2] b^T (+ bias): a1 = P3 [m][k1], a2 = P3 [m][k2] (nullable, k2 = 0), b = P3 [n][ceil16(k1) + k2]
// (the K blocks of the second segment follow the ceil(k1 / 16) blocks of the first in every row of b)
static int gemm_p3_nt_impl(const void* a1, int64_t lda1, int64_t k1, const void* a2, int64_t lda2, int64_t k2, const void* b,
                           int64_t ldb, const float* bias, int64_t bias_cols, float* c, int64_t ldc, int64_t m, int64_t n,
                           int relu, int accumulate, void* stream, const int32_t* a_rows, int64_t n_res_rows, bool rows_both = false) {
    if (m < 0 || n < 0 || k1 <= 0 || k2 < 0 || m > INT32_MAX || n > INT32_MAX || k1 + k2 > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt: bad sizes");
    if (m == 0 || n == 0) return GTE_OK;
    if (!a1 || !b || !c || (k2 > 0 && !a2)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt: null pointer");
    const int64_t kb1 = p3::blocks(k1), kb2 = p3::blocks(k2);
    const P3Strides sb = p3_strides(ldb);
    if (lda1 < kb1 * 96 || (k2 > 0 && lda2 < kb2 * 96) || !p3_strides_ok(ldb, n, 16 * (kb1 + kb2)) || ldc < n)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt: leading dimension too small");
    if (lda1 >= (1 << 22) || lda2 >= (1 << 22) || sb.ld >= (1 << 22) || (m + 256) * ldc * 4 >= ((int64_t)1 << 31))
        return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt: row strides must be < 4 MB and the output < 2 GB");
    P3Gemm p = {};
    p.A1 = (const char*)a1; p.lda1 = lda1; p.KB1 = (int)kb1;
    p.A2 = k2 > 0 ? (const char*)a2 : nullptr; p.lda2 = lda2; p.KB2 = (int)kb2;
    p.B = (const char*)b; p.ldb = sb.ld; p.C = c; p.ldc = ldc; p.bias = bias; p.bias_cols = (int)bias_cols;
    p.bsa1 = p.bsa2 = 96; p.bsb = sb.bs;
    p.M = (int)m; p.N = (int)n; p.relu = relu; p.accumulate = accumulate; p.splits = 1;
    if (a_rows) {
        if (k2 > 0 && !rows_both) return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt_rows: one K segment only (two: gte_gemm_p3_nt_rows2)");
        if (n_res_rows <= 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_rows: empty resident image");
        p.rowsA = a_rows; p.res_bytes = n_res_rows * lda1;
        p.rows64 = rows64_needed(p.res_bytes);
        if (rows_both) {
            p.rows_both = 1; p.res_bytes2 = n_res_rows * lda2;
            if (rows64_needed(p.res_bytes2)) p.rows64 = 1;     // (64-bit per-lane addresses for both images)
        }
    }
    return launch_nt(p, gte::as_stream(stream));
}

extern "C" int gte_gemm_p3_nt(const void* a1, int64_t lda1, int64_t k1, const void* a2, int64_t lda2, int64_t k2, const void* b,
                              int64_t ldb, const float* bias, int64_t bias_cols, float* c, int64_t ldc, int64_t m, int64_t n,
                              int relu, int accumulate, void* stream) {
    return gemm_p3_nt_impl(a1, lda1, k1, a2, lda2, k2, b, ldb, bias, bias_cols, c, ldc, m, n, relu, accumulate, stream, nullptr, 0);
}

// c[m, n] (+)= A b^T with A = the rows a_rows[0 .. m) of a RESIDENT P3 image a_res [n_res_rows][k]: the input layer's forward
// transform straight from the resident features (no per-batch copy of the rows).  One K segment.  Images of 4 GB or more are
// read through 64-bit per-lane addresses (the loader-wave kernels, 128 / 192 x 256 tiles).
extern "C" int gte_gemm_p3_nt_rows(const void* a_res, int64_t ldpa, int64_t k, const int32_t* a_rows, int64_t n_res_rows, const void* b,
                                   int64_t ldpb, const float* bias, int64_t bias_cols, float* c, int64_t ldc, int64_t m, int64_t n,
                                   int relu, int accumulate, void* stream) {
    if (!a_rows) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_rows: null row map");
    return gemm_p3_nt_impl(a_res, ldpa, k, nullptr, 0, 0, b, ldpb, bias, bias_cols, c, ldc, m, n, relu, accumulate, stream, a_rows, n_res_rows);
}

// c[m, n] (+)= [A | A2] b^T with A / A2 = the rows a_rows[0 .. m) of TWO resident P3 images of n_res_rows rows and k columns each
// (the input features and their cached mean aggregate: an aggregate-first input layer without any per-batch operand preparation);
// b = P3 [n][2 ceil16(k)], the second K segment at block ceil(k / 16).  Images below 4 GB.
extern "C" int gte_gemm_p3_nt_rows2(const void* a_res, int64_t ldpa, const void* a2_res, int64_t ldpa2, int64_t k, const int32_t* a_rows,
                                    int64_t n_res_rows, const void* b, int64_t ldpb, const float* bias, int64_t bias_cols, float* c,
                                    int64_t ldc, int64_t m, int64_t n, int relu, int accumulate, void* stream) {
    if (!a_rows || !a2_res) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_rows2: null row map / second image");
    return gemm_p3_nt_impl(a_res, ldpa, k, a2_res, ldpa2, k, b, ldpb, bias, bias_cols, c, ldc, m, n, relu, accumulate, stream, a_rows,
                           n_res_rows, true);
}

namespace {
template <int WM, int WN, int TM, int TN, int NBUF, int WGS>
void launch_ring(const P3Gemm& p, hipStream_t s) {
    constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NI = ((BM + BN) * 96 / 1024 + NW - 1) / NW, shm = NBUF * NI * NW * 1024;
    static bool configured = false;
    if (!configured) {
        GTE_SET_LDS((gemm_p3_nt_ring_kernel<WM, WN, TM, TN, NBUF, WGS>), shm);
        configured = true;
    }
    const dim3 grid((unsigned)(gte::ceil_div(p.M, BM) * gte::ceil_div(p.N, BN)));
    hipLaunchKernelGGL((gemm_p3_nt_ring_kernel<WM, WN, TM, TN, NBUF, WGS>), grid, dim3(NW * 64), shm, s, p);
}
template <int WM, int WN, int TM, int TN, int NL, bool BIG = false>
void launch_lw(const P3Gemm& p, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NI = ((BM + BN) * 96 / 1024 + NL - 1) / NL, shm = 3 * NI * NL * 1024;
    static bool configured = false;
    if (!configured) {
        GTE_SET_LDS((gemm_p3_nt_lw_kernel<WM, WN, TM, TN, NL, 0, BIG>), shm);
        configured = true;
    }
    const dim3 grid((unsigned)(gte::ceil_div(p.M, BM) * gte::ceil_div(p.N, BN)));
    hipLaunchKernelGGL((gemm_p3_nt_lw_kernel<WM, WN, TM, TN, NL, 0, BIG>), grid, dim3((WM * WN + NL) * 64), shm, s, p);
}
template <int WM, int WN, int TM, int TN, int NL, int LNB = 1, bool BIG = false>
void launch_lw_lnb(const P3Gemm& p, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NI = ((BM + BN) * 96 / 1024 + NL - 1) / NL, shm = 3 * NI * NL * 1024;
    static_assert(shm >= TM * 32 * 256 * 4 && shm >= WM * WN * 3 * 256 * 4, "the epilogue's row slice lives in the stage images");
    static bool configured = false;
    if (!configured) {
        GTE_SET_LDS((gemm_p3_nt_lw_kernel<WM, WN, TM, TN, NL, LNB, BIG>), shm);
        configured = true;
    }
    const dim3 grid((unsigned)gte::ceil_div(p.M, BM));
    hipLaunchKernelGGL((gemm_p3_nt_lw_kernel<WM, WN, TM, TN, NL, LNB, BIG>), grid, dim3((WM * WN + NL) * 64), shm, s, p);
}
// The block-major-weights kernel takes a product when the weights ARE block-major, the rows are reached through 32-bit offsets, one
// column of tiles covers the output, and cutting the K blocks into slots of four pads the K loop by at most 1 / 12 ...
inline bool sq_applies(const P3Gemm& p) {
    if (p.ldb != 96 || p.bsb < 96 || p.rows64 || p.N > 256 || p.bsa1 != 96 || p.bsa2 != 96) return false;
    // ... and K is at least 32 blocks deep: below that the slots' longer pipeline fill (four K blocks before the first MFMA) costs what
    // the loop gains -- dX + LayerNorm backward at K = 2 x 13 / 2 x 14 blocks: 42.9 -> 44.5 / 44.2 -> 45.2 us; at 32 blocks 57.3 -> 50.4,
    // at 52 blocks 56.6 -> 52.0, at 104 blocks 122 -> 108 (profiles/r06/sequence_*.txt)
    const int T = p.KB1 + p.KB2, S4 = 4 * ((T + 3) / 4);
    return T >= 32 && (S4 - T) * 12 <= T;
}
// ... a PLAIN product (no LayerNorm epilogue) additionally wants its one column of tiles more than half full
inline bool sq_plain_wanted(const P3Gemm& p) { return p.N > 128; }
// the block-major-weights kernel (gemm_p3_nt_sq_kernel): (32 TM) x 256 tiles
template <int TM, int NL, int LNB>
void launch_sq(const P3Gemm& p, hipStream_t s) {
    constexpr int BM = TM * 32, shm = 3 * 4 * BM * 96 + 1024;
    static_assert(LNB == 0 || (shm >= BM * 256 * 4 && shm >= 8 * 3 * 256 * 4), "the epilogue's row slice lives in the slots");
    static bool configured = false;
    if (!configured) {
        GTE_SET_LDS((gemm_p3_nt_sq_kernel<TM, NL, LNB>), shm);
        configured = true;
    }
    const dim3 grid((unsigned)(gte::ceil_div(p.M, BM) * gte::ceil_div(p.N, 256)));
    hipLaunchKernelGGL((gemm_p3_nt_sq_kernel<TM, NL, LNB>), grid, dim3((8 + NL) * 64), shm, s, p);
}
// Configurations (measured on the step's shapes, profiles/r03/gemm_p3.md): the kernel runs at the chip's power limit
// (~1.3 PF bf16 whatever the tile), so what matters is ONE balanced round of tiles: the row tile is the smallest of
// 128 / 160 / 192 / 224 / 256 that covers M with at most one tile per CU; N is cut in 256-column tiles.
struct NtCfg { int id, bm, bn, wgs; };
// (Round 5 measured column tiles of 160 / 192 / 224 -- 1 x WN waves of (32 TM) x 32 -- for the output widths of the reference's
// scaled runs, 2 x 96 ... 2 x 224 columns, which run 14 ... 78 % padding through the matrix pipe on 256-wide tiles: no gain, 1 - 3 %
// slower on five of six shapes (profiles/r05/nt_tile_widths.txt): these launches are bound by the operand stream into LDS, the
// padded MFMAs are free, and the 2 x 4 wave layout with loader waves hides more latency.  Removed again.)
constexpr NtCfg kNtCfg[] = {{0, 64, 128, 3}, {1, 128, 128, 2}, {2, 128, 256, 1}, {3, 160, 256, 1}, {4, 192, 256, 1}, {5, 224, 256, 1},
                            {6, 256, 256, 1}};
// forced tile configuration: -1 = the chooser; gte_gemm_p3_set_nt_cfg (tests force every configuration in ONE process; until round 5
// this was a static getenv read once -- the per-configuration tests that set the variable after the first GEMM of the process ran
// the chooser's pick every time) or GTE_P3_NT_CFG at the first call
thread_local int g_nt_cfg = -2;
int nt_choose(const P3Gemm& p) {
    if (g_nt_cfg == -2) g_nt_cfg = GTE_MEASURE_INT("GTE_P3_NT_CFG", -1);
    const int forced = g_nt_cfg;
    if (forced >= 0 && forced <= 7) return forced;
    const int cus = gte::device_props().cus;
    double best = 1e30;
    int bi = 0;
    for (const NtCfg& c : kNtCfg) {
        if (p.N <= 128 && c.bn > 128) continue;
        const int64_t tiles = gte::ceil_div(p.M, c.bm) * gte::ceil_div(p.N, c.bn);
        const int64_t rounds = gte::ceil_div(tiles, (int64_t)cus * c.wgs);
        // makespan in units of tile area; small tiles pay for their operand traffic (load-bound below ~128 x 256)
        const double penalty = c.bm * c.bn >= 128 * 256 ? 1.0 : (c.bm * c.bn >= 128 * 128 ? 1.15 : 1.35);
        const double t = (double)rounds * c.wgs * c.bm * c.bn * penalty;
        if (t < best) { best = t; bi = c.id; }
    }
    return bi;
}
// the smallest of 32 / 64 / 96 rows that covers m in ONE round of workgroups, else 128
inline int one_round_row_tile(int64_t m, int64_t cus, int min_tile = 32) {
    for (int t = 32; t <= 96; t += 32)
        if (t >= min_tile && gte::ceil_div(m, t) <= cus) return t;
    return 128;
}
int lnb_row_tile(int64_t m);
int launch_nt(const P3Gemm& p, hipStream_t s) {
    if (p.rowsA && p.rows64) {
        // a row map into an image of 4 GB or more: the two loader-wave tiles, whichever makes the shorter schedule
        const int cus = gte::device_props().cus;
        const int64_t r128 = gte::ceil_div(gte::ceil_div(p.M, 128) * gte::ceil_div(p.N, 256), cus) * 128;
        const int64_t r192 = gte::ceil_div(gte::ceil_div(p.M, 192) * gte::ceil_div(p.N, 256), cus) * 192;
        if (r192 < r128) launch_lw<2, 4, 3, 2, 4, true>(p, s); else launch_lw<2, 4, 2, 2, 4, true>(p, s);
        return gte::check_launch("gemm_p3_nt_rows");
    }
    if (g_nt_cfg == -2) g_nt_cfg = GTE_MEASURE_INT("GTE_P3_NT_CFG", -1);
    if (g_nt_cfg < 0 && sq_applies(p) && sq_plain_wanted(p)) {
        // block-major weights and one column of tiles more than half full: the block-major-weights kernel, on the smallest row tile
        // that covers M in one round (the narrow-width input GEMMs of the scaled runs: 2 x 96 ... 2 x 128 columns over K = 781 / 831)
        const int bm = lnb_row_tile(p.M);
        if (bm == 32) launch_sq<1, 4, 0>(p, s);
        else if (bm == 64) launch_sq<2, 4, 0>(p, s);
        else if (bm == 96) launch_sq<3, 4, 0>(p, s);
        else launch_sq<4, 4, 0>(p, s);
        return gte::check_launch("gemm_p3_nt");
    }
    int cfg = nt_choose(p);
    switch (cfg) {
        case 7: launch_ring<2, 4, 3, 2, 3, 1>(p, s); break;     // measurement: 192 x 256 ring (no loader waves)
        case 0: launch_ring<2, 2, 1, 2, 2, 3>(p, s); break;    //  64 x 128, three workgroups per CU
        case 1: launch_ring<2, 2, 2, 2, 3, 2>(p, s); break;    // 128 x 128, two
        case 2: launch_lw<2, 4, 2, 2, 4>(p, s); break;         // 128 x 256: 8 compute + 4 loader waves
        case 3: launch_ring<1, 8, 5, 1, 3, 1>(p, s); break;    // 160 x 256: 1 x 8 waves of 160 x 32
        case 4: launch_lw<2, 4, 3, 2, 4>(p, s); break;         // 192 x 256: 8 compute + 4 loader waves
        case 5: launch_ring<1, 8, 7, 1, 3, 1>(p, s); break;    // 224 x 256
        default: launch_ring<2, 4, 4, 2, 3, 1>(p, s); break;   // 256 x 256
    }
    return gte::check_launch("gemm_p3_nt");
}
}  // namespace

// Which kernel family an NT product with ONE column of tiles takes (pure host logic: no device call when `cus` > 0).  Returns 1 for
// the block-major-weights kernel (gemm_p3_nt_sq_kernel), 0 for the loader-wave / ring kernels; *row_tile = rows of its tile where
// the launch picks one by the one-round rule (the LayerNorm-epilogue launches and the narrow plain products), else 0.
extern "C" int gte_gemm_p3_nt_plan(int64_t m, int64_t n, int64_t k1, int64_t k2, int weights_block_major, int epilogue, int cus,
                                   int* row_tile) {
    if (row_tile) *row_tile = 0;
    if (m <= 0 || n <= 0 || k1 <= 0 || k2 < 0 || (epilogue != 0 && epilogue != 1 && epilogue != 3 && epilogue != 4))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_plan: bad arguments (epilogue: 0 plain, 1 / 3 LayerNorm backward, 4 LayerNorm forward)");
    P3Gemm p = {};
    p.M = (int)m; p.N = (int)n; p.KB1 = (int)p3::blocks(k1); p.KB2 = (int)p3::blocks(k2);
    p.bsa1 = p.bsa2 = 96;
    p.ldb = weights_block_major ? 96 : (p.KB1 + p.KB2) * 96; p.bsb = weights_block_major ? gte::round_up(n, 16) * 96 : 96;
    const int bm = one_round_row_tile(m, cus > 0 ? cus : gte::device_props().cus);
    const bool sq = sq_applies(p) && (epilogue != 0 || sq_plain_wanted(p));
    if (row_tile && n <= 256 && (epilogue != 0 || sq)) *row_tile = bm;
    return sq ? 1 : 0;
}

extern "C" int gte_gemm_p3_set_nt_cfg(int cfg) {
    if (cfg < -1 || cfg > 7) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_set_nt_cfg: -1 (chooser) or a configuration id 0 ... 7");
    g_nt_cfg = cfg;
    return GTE_OK;
}

// ---- the NT product with the LayerNorm(+ReLU) backward of the layer below as its epilogue ------------------------------
namespace {
// Row tile of the fused dX + LayerNorm-backward launch: 128.  (A 192-row instantiation served m > 128 x #CUs until round 3: its
// epilogue's row slice + 96 accumulator registers did not fit the 170-register budget of three waves per SIMD -- 140 bytes of
// scratch per lane.  Batches beyond one round of 128-row tiles run several rounds of the same kernel.)
// Round 5: 96 rows (1 x 8 waves of 96 x 32) while that covers m in ONE round of tiles -- 24 317 rows of the headline batch are 190
// tiles of 128 rows on 256 CUs, a quarter of the chip idle for the whole launch; 254 tiles of 96 rows fill it.
// ... and 64 / 32 rows (1 x 8 waves of 64 x 32 / 32 x 32) for batches that leave CUs idle even then: the launch lasts as long
// as ONE tile's K loop whatever the number of tiles in the round (L0 forward, K = 1 662: 106 us on 50 tiles of 128 rows, 109 us
// on 95, 124 us on 202 -- profiles/r05/sequence_pages*.txt), so a small batch wants the smallest tile that still is one round.
thread_local int g_ln_rows = -1;        // forced row tile of the LayerNorm-epilogue launches: 0 = the chooser (gte_gemm_p3_set_ln_rows)
int lnb_row_tile(int64_t m) {
    if (g_ln_rows < 0) g_ln_rows = GTE_MEASURE_INT("GTE_P3_LN_ROWS", 0);        // (32 / 64 / 96 / 128)
    const int forced = g_ln_rows;
    if (forced == 32 || forced == 64 || forced == 96 || forced == 128) return forced;
    static const int min_tile = GTE_MEASURE_INT("GTE_P3_LN_MIN_ROWS", 32);
    return one_round_row_tile(m, gte::device_props().cus, min_tile);
}
// out[j] = sum_k part[k * stride + j]   (only when no fold deferral is open)
__global__ void __launch_bounds__(256)
p3_colsum_fold_kernel(const float* __restrict__ part, long long stride, int count, int n, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int k = 0; k < count; ++k) s += part[k * stride + j];
    out[j] = s;
}
}
extern "C" int gte_gemm_p3_set_ln_rows(int rows) {
    if (rows != 0 && rows != 32 && rows != 64 && rows != 96 && rows != 128)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_set_ln_rows: 0 (chooser), 32, 64, 96 or 128");
    g_ln_rows = rows;
    return GTE_OK;
}
// any width up to 256 (a workgroup's tile holds whole rows); n % 4 != 0 needs rows of z / dz padded to a multiple of 4 floats
extern "C" int gte_gemm_p3_nt_ln_bwd_supported(int64_t n) { return (n >= 1 && n <= 256) ? 1 : 0; }
extern "C" int64_t gte_gemm_p3_nt_ln_bwd_workspace_bytes(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 256;
    return gte::round_up(gte::ceil_div(m, 32) * 3 * n * 4, 256);       // (the smallest row tile: covers every choice, for every m' <= m)
}
// dy = [a1 | a2] b^T (m x n, n <= 256: a workgroup's tile holds whole rows) is NOT stored: the workgroup that computed a row
// block runs the LayerNorm(+ReLU) backward of those rows on it -- dz = LN'(z)(mask . dy) as fp32 (feeds the transpose
// aggregation) and as a P3 image (dzp3 nullable; feeds the layer's dW / dX GEMMs) -- and the column sums dgamma / dbeta / dbias
// (each nullable) join the fold deferral.  Same arithmetic as gte_gemm_p3_nt + gte_ln_relu_bwd_p3: dz is bit-identical.  dz itself is
// nullable when dzp3 is given (the layer below reads the image only: an input layer on the cached aggregate has no transpose
// aggregation, its dW = dz^T [x | ahn] takes the image -- 25 MB per step less at 24 k x 256).
extern "C" int gte_gemm_p3_nt_ln_bwd(const void* a1, int64_t lda1, int64_t k1, const void* a2, int64_t lda2, int64_t k2, const void* b,
                                     int64_t ldb, const float* z, int64_t ldz, const float* stats, const float* gamma,
                                     const float* beta, int relu, float* dz, int64_t lddz, void* dzp3, int64_t ldp3, float* dgamma,
                                     float* dbeta, float* dbias, int64_t m, int64_t n, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
    if (m < 0 || n <= 0 || k1 <= 0 || k2 < 0 || m > INT32_MAX || k1 > INT32_MAX || k2 > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_ln_bwd: bad sizes");
    if (!gte_gemm_p3_nt_ln_bwd_supported(n)) return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt_ln_bwd: needs n <= 256");
    if (m == 0) return GTE_OK;
    if (!a1 || !b || !z || !stats || !gamma || !beta || (!dz && !dzp3) || !workspace || (k2 > 0 && !a2))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_ln_bwd: null pointer");
    const int64_t kb1 = p3::blocks(k1), kb2 = k2 > 0 ? p3::blocks(k2) : 0;
    const int64_t n4 = gte::round_up(n, 4);
    const P3Strides sb = p3_strides(ldb);
    if (lda1 < p3::row_bytes(k1) || (k2 > 0 && lda2 < p3::row_bytes(k2)) || !p3_strides_ok(ldb, n, 16 * (kb1 + kb2)) || ldz < n4 ||
        (dz && lddz < n4) || (dzp3 && (ldp3 < p3::row_bytes(n) || ldp3 % 16 != 0)))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_ln_bwd: leading dimension too small");
    if (lda1 >= (1 << 23) || lda2 >= (1 << 23) || sb.ld >= (1 << 23) || (m + 256) * lda1 >= ((int64_t)1 << 31) ||
        (k2 > 0 && (m + 256) * lda2 >= ((int64_t)1 << 31)))
        return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt_ln_bwd: operand images must be < 2 GB with row strides < 8 MB");
    const int64_t need = gte_gemm_p3_nt_ln_bwd_workspace_bytes(m, n);
    if (workspace_bytes < need)
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "gemm_p3_nt_ln_bwd: needs %lld workspace bytes, got %lld", (long long)need,
                         (long long)workspace_bytes);
    P3Gemm p = {};
    p.A1 = reinterpret_cast<const char*>(a1); p.lda1 = lda1; p.KB1 = (int)kb1;
    p.A2 = k2 > 0 ? reinterpret_cast<const char*>(a2) : nullptr; p.lda2 = lda2; p.KB2 = (int)kb2;
    p.B = reinterpret_cast<const char*>(b); p.ldb = sb.ld;
    p.bsa1 = p.bsa2 = 96; p.bsb = sb.bs;
    p.M = (int)m; p.N = (int)n; p.splits = 1;
    p.ln_z = z; p.ln_ldz = ldz; p.ln_stats = stats; p.ln_gamma = gamma; p.ln_beta = beta; p.ln_relu = relu;
    p.ln_dz = dz; p.ln_lddz = lddz; p.ln_dzp3 = reinterpret_cast<char*>(dzp3); p.ln_ldp3 = ldp3;
    p.ln_part = reinterpret_cast<float*>(workspace);
    hipStream_t s = gte::as_stream(stream);
    const int bm = lnb_row_tile(m);
    // (column tiles of 128 / 192 for hidden widths up to 128 / 192 were measured in round 5 and removed: 27 -> 30 us at 96 columns)
    if (sq_applies(p)) {                                  // block-major weights: fragments straight to registers, 64-deep A slots
        if (n % 16 == 0) {
            if (bm == 32) launch_sq<1, 4, 1>(p, s);
            else if (bm == 64) launch_sq<2, 4, 1>(p, s);
            else if (bm == 96) launch_sq<3, 4, 1>(p, s);
            else launch_sq<4, 4, 1>(p, s);
        } else {
            if (bm == 32) launch_sq<1, 4, 3>(p, s);
            else if (bm == 64) launch_sq<2, 4, 3>(p, s);
            else if (bm == 96) launch_sq<3, 4, 3>(p, s);
            else launch_sq<4, 4, 3>(p, s);
        }
    } else if (n % 16 == 0) {
        if (bm == 32) launch_lw_lnb<1, 8, 1, 1, 4>(p, s);
        else if (bm == 64) launch_lw_lnb<1, 8, 2, 1, 4>(p, s);
        else if (bm == 96) launch_lw_lnb<1, 8, 3, 1, 4>(p, s);
        else launch_lw_lnb<2, 4, 2, 2, 4>(p, s);
    } else {                                              // per-element validity, zero image columns up to the next multiple of 16
        if (bm == 32) launch_lw_lnb<1, 8, 1, 1, 4, 3>(p, s);
        else if (bm == 64) launch_lw_lnb<1, 8, 2, 1, 4, 3>(p, s);
        else if (bm == 96) launch_lw_lnb<1, 8, 3, 1, 4, 3>(p, s);
        else launch_lw_lnb<2, 4, 2, 2, 4, 3>(p, s);
    }
    int rc = gte::check_launch("gemm_p3_nt_ln_bwd");
    if (rc != GTE_OK) return rc;
    const int nb = (int)gte::ceil_div(m, bm);
    if (gte::defer_fold(p.ln_part, 3 * n, nb, 1, (int)n, dgamma, n)) {
        gte::defer_fold(p.ln_part + n, 3 * n, nb, 1, (int)n, dbeta, n);
        gte::defer_fold(p.ln_part + 2 * n, 3 * n, nb, 1, (int)n, dbias, n);
        return GTE_OK;
    }
    float* outs[3] = {dgamma, dbeta, dbias};
    for (int i = 0; i < 3; ++i)
        if (outs[i])
            hipLaunchKernelGGL(p3_colsum_fold_kernel, dim3((unsigned)gte::ceil_div(n, 256)), dim3(256), 0, s, p.ln_part + i * n,
                               (long long)3 * n, nb, (int)n, outs[i]);
    return gte::check_launch("gemm_p3_nt_ln_bwd fold");
}

// ---- the NT product with LayerNorm(+ReLU) FORWARD as its epilogue ------------------------------------------------------------
// z = [a1 | a2] b^T + bias (m x n, n <= 256: a workgroup's tile holds whole rows) is written as fp32 (the operand of the layer's
// LayerNorm backward), and the workgroup that computed a row block normalises it: stats (mean, rstd), y = relu?(LN(z)) as fp32
// (nullable) and as a P3 image (nullable).  Bit-identical to gte_gemm_p3_nt + gte_ln_relu_fwd_p3 (one launch, the z read-back and
// the LayerNorm launch gone).  Rows of z / y padded to a multiple of 4 floats.
static int gemm_p3_nt_ln_fwd_impl(const void* a1, int64_t lda1, int64_t k1, const void* a2, int64_t lda2, int64_t k2, const int32_t* a_rows,
                                  int64_t n_res_rows, const void* b, int64_t ldb, const float* bias, const float* gamma,
                                  const float* beta, float eps, int relu, float* z, int64_t ldz, float* y, int64_t ldy, void* yp3,
                                  int64_t ldyp3, float* stats, int64_t m, int64_t n, void* stream) {
    if (m < 0 || n <= 0 || n > 256 || k1 <= 0 || k2 < 0 || m > INT32_MAX || k1 > INT32_MAX || k2 > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_ln_fwd: bad sizes (n <= 256)");
    if (m == 0) return GTE_OK;
    if (!a1 || !b || !z || !gamma || !beta || (!y && !yp3) || (k2 > 0 && !a2))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_ln_fwd: null pointer");
    const int64_t kb1 = p3::blocks(k1), kb2 = k2 > 0 ? p3::blocks(k2) : 0, n4 = gte::round_up(n, 4);
    const P3Strides sb = p3_strides(ldb);
    if (lda1 < kb1 * 96 || (k2 > 0 && lda2 < kb2 * 96) || !p3_strides_ok(ldb, n, 16 * (kb1 + kb2)) || ldz < n4 || (y && ldy < n4) ||
        (yp3 && (ldyp3 < p3::row_bytes(n) || ldyp3 % 16 != 0)))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_ln_fwd: leading dimension too small (rows of z / y padded to 4 floats)");
    if (lda1 >= (1 << 22) || lda2 >= (1 << 22) || sb.ld >= (1 << 22))
        return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt_ln_fwd: row strides must be < 4 MB");
    P3Gemm p = {};
    p.A1 = (const char*)a1; p.lda1 = lda1; p.KB1 = (int)kb1;
    p.A2 = k2 > 0 ? (const char*)a2 : nullptr; p.lda2 = lda2; p.KB2 = (int)kb2;
    p.B = (const char*)b; p.ldb = sb.ld; p.C = z; p.ldc = ldz; p.bias = bias;
    p.bsa1 = p.bsa2 = 96; p.bsb = sb.bs;
    p.M = (int)m; p.N = (int)n; p.splits = 1;
    p.ln_gamma = gamma; p.ln_beta = beta; p.ln_relu = relu;
    p.lnf_y = y; p.lnf_ldy = ldy; p.lnf_yp3 = reinterpret_cast<char*>(yp3); p.lnf_ldp = ldyp3; p.lnf_stats = stats; p.lnf_eps = eps;
    if (a_rows) {
        if (n_res_rows <= 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_ln_fwd: empty resident image");
        p.rowsA = a_rows; p.res_bytes = n_res_rows * lda1;
        if (k2 > 0) { p.rows_both = 1; p.res_bytes2 = n_res_rows * lda2; }
        p.rows64 = (rows64_needed(p.res_bytes) || (k2 > 0 && rows64_needed(p.res_bytes2))) ? 1 : 0;
    }
    if (p.rows64)                                          // images of 4 GB or more: 64-bit per-lane addresses (no range check: the
        launch_lw_lnb<2, 4, 2, 2, 4, 4, true>(p, gte::as_stream(stream));                    // rows past the tile re-read its first row)
    else if (sq_applies(p)) {                              // block-major weights: fragments straight to registers, 64-deep A slots
        const int bm = lnb_row_tile(m);
        if (bm == 32) launch_sq<1, 4, 4>(p, gte::as_stream(stream));
        else if (bm == 64) launch_sq<2, 4, 4>(p, gte::as_stream(stream));
        else if (bm == 96) launch_sq<3, 4, 4>(p, gte::as_stream(stream));
        else launch_sq<4, 4, 4>(p, gte::as_stream(stream));
    } else if (lnb_row_tile(m) == 32) launch_lw_lnb<1, 8, 1, 1, 4, 4>(p, gte::as_stream(stream));
    else if (lnb_row_tile(m) == 64) launch_lw_lnb<1, 8, 2, 1, 4, 4>(p, gte::as_stream(stream));
    else if (lnb_row_tile(m) == 96) launch_lw_lnb<1, 8, 3, 1, 4, 4>(p, gte::as_stream(stream));
    else launch_lw_lnb<2, 4, 2, 2, 4, 4>(p, gte::as_stream(stream));
    return gte::check_launch("gemm_p3_nt_ln_fwd");
}
extern "C" int gte_gemm_p3_nt_ln_fwd_supported(int64_t n) { return (n >= 1 && n <= 256) ? 1 : 0; }
extern "C" int gte_gemm_p3_nt_ln_fwd(const void* a1, int64_t lda1, int64_t k1, const void* a2, int64_t lda2, int64_t k2, const void* b,
                                     int64_t ldb, const float* bias, const float* gamma, const float* beta, float eps, int relu, float* z,
                                     int64_t ldz, float* y, int64_t ldy, void* yp3, int64_t ldyp3, float* stats, int64_t m, int64_t n,
                                     void* stream) {
    return gemm_p3_nt_ln_fwd_impl(a1, lda1, k1, a2, lda2, k2, nullptr, 0, b, ldb, bias, gamma, beta, eps, relu, z, ldz, y, ldy, yp3, ldyp3,
                                  stats, m, n, stream);
}
// ... with [A | A2] = the rows a_rows[0 .. m) of two resident images of k columns each (gte_gemm_p3_nt_rows2): the whole forward of
// an input layer on its features and their cached mean aggregate in ONE launch
extern "C" int gte_gemm_p3_nt_rows2_ln_fwd(const void* a_res, int64_t ldpa, const void* a2_res, int64_t ldpa2, int64_t k,
                                           const int32_t* a_rows, int64_t n_res_rows, const void* b, int64_t ldb, const float* bias,
                                           const float* gamma, const float* beta, float eps, int relu, float* z, int64_t ldz, float* y,
                                           int64_t ldy, void* yp3, int64_t ldyp3, float* stats, int64_t m, int64_t n, void* stream) {
    if (!a_rows || !a2_res) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_rows2_ln_fwd: null row map / second image");
    return gemm_p3_nt_ln_fwd_impl(a_res, ldpa, k, a2_res, ldpa2, k, a_rows, n_res_rows, b, ldb, bias, gamma, beta, eps, relu, z, ldz, y, ldy,
                                  yp3, ldyp3, stats, m, n, stream);
}

// ---- ... with the WHOLE backward of a short-input layer below (gte_sage_smallk_bwd) as its epilogue --------------------------
extern "C" int gte_gemm_p3_nt_smallk_bwd_supported(int64_t k_total, int64_t n) {
    return (n >= 4 && n <= 256 && n % 4 == 0 && k_total >= 1 && k_total <= 28) ? 1 : 0;
}
extern "C" int64_t gte_gemm_p3_nt_smallk_bwd_workspace_bytes(int64_t m, int64_t k_total, int64_t n) {
    if (m <= 0 || n <= 0 || k_total <= 0) return 256;
    return gte::round_up(gte::ceil_div(m, 128) * n * (k_total + 3) * 4, 256);        // (always the 128-row tile)
}
// dy = [a1 | a2] b^T (m x n, n <= 256) is the gradient w.r.t. the output of a short-input INPUT layer y = relu?(LN([x | ahn] W^T +
// bias)); it is not stored: the workgroup that computed a row block recomputes z from the k1 + k2 <= 28 inputs per row, runs the
// LayerNorm(+ReLU) backward and accumulates dW = dz^T [x | ahn]; dW / dbias / dgamma / dbeta join the fold deferral.  Replaces
// gte_gemm_p3_nt + gte_sage_smallk_bwd (one launch, the 2 m n 4 bytes of the dy round trip).
extern "C" int gte_gemm_p3_nt_smallk_bwd(const void* a1, int64_t lda1, int64_t kg1, const void* a2, int64_t lda2, int64_t kg2,
                                         const void* b, int64_t ldb, const float* x, int64_t ldx, int64_t k1, const float* ahn,
                                         int64_t ldahn, int64_t k2, const float* W, int64_t ldw, const float* bias, const float* gamma,
                                         const float* beta, const float* stats, int relu, float* dW, int64_t lddw, float* dbias,
                                         float* dgamma, float* dbeta, int64_t m, int64_t n, void* workspace, int64_t workspace_bytes,
                                         void* stream) {
    if (m < 0 || n <= 0 || kg1 <= 0 || kg2 < 0 || k1 <= 0 || k2 < 0 || m > INT32_MAX || kg1 > INT32_MAX || kg2 > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_smallk_bwd: bad sizes");
    const int64_t K = k1 + k2;
    if (!gte_gemm_p3_nt_smallk_bwd_supported(K, n))
        return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt_smallk_bwd: needs n <= 256, n %% 4 == 0, k1 + k2 <= 28");
    if (m == 0) return GTE_OK;
    if (!a1 || !b || !x || (k2 > 0 && !ahn) || !W || !bias || !gamma || !beta || !stats || !dW || !workspace || (kg2 > 0 && !a2))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_smallk_bwd: null pointer");
    const int64_t kb1 = p3::blocks(kg1), kb2 = kg2 > 0 ? p3::blocks(kg2) : 0;
    const P3Strides sb = p3_strides(ldb);
    if (lda1 < p3::row_bytes(kg1) || (kg2 > 0 && lda2 < p3::row_bytes(kg2)) || !p3_strides_ok(ldb, n, 16 * (kb1 + kb2)) || ldx < k1 || (k2 > 0 && ldahn < k2) ||
        ldw < K || lddw < K)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_nt_smallk_bwd: leading dimension too small");
    if (lda1 >= (1 << 23) || lda2 >= (1 << 23) || sb.ld >= (1 << 23) || (m + 256) * lda1 >= ((int64_t)1 << 31) ||
        (kg2 > 0 && (m + 256) * lda2 >= ((int64_t)1 << 31)))
        return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt_smallk_bwd: operand images must be < 2 GB with row strides < 8 MB");
    const int64_t need = gte_gemm_p3_nt_smallk_bwd_workspace_bytes(m, K, n);
    if (workspace_bytes < need)
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "gemm_p3_nt_smallk_bwd: needs %lld workspace bytes, got %lld", (long long)need,
                         (long long)workspace_bytes);
    const int bm = 128;       // (the 192-row tile's stage images are 2.5 KB short of the row slice + W^T + inputs)
    const int nb = (int)gte::ceil_div(m, bm);
    P3Gemm p = {};
    p.A1 = reinterpret_cast<const char*>(a1); p.lda1 = lda1; p.KB1 = (int)kb1;
    p.A2 = kg2 > 0 ? reinterpret_cast<const char*>(a2) : nullptr; p.lda2 = lda2; p.KB2 = (int)kb2;
    p.B = reinterpret_cast<const char*>(b); p.ldb = sb.ld;
    p.bsa1 = p.bsa2 = 96; p.bsb = sb.bs;
    p.M = (int)m; p.N = (int)n; p.splits = 1;
    p.ln_stats = stats; p.ln_gamma = gamma; p.ln_beta = beta; p.ln_relu = relu;
    p.sk_x = x; p.sk_ldx = ldx; p.sk_k1 = (int)k1; p.sk_ahn = k2 > 0 ? ahn : nullptr; p.sk_ldahn = ldahn; p.sk_k2 = (int)k2;
    p.sk_W = W; p.sk_ldw = ldw; p.sk_bias = bias;
    p.sk_part_dw = reinterpret_cast<float*>(workspace);
    p.ln_part = p.sk_part_dw + (int64_t)nb * n * K;
    hipStream_t s = gte::as_stream(stream);
    {
        constexpr int shm = 160 * 1024;                             // the 128 x 256 ring tile (3 x 40 KB) + room for the whole fp32 tile
        static bool configured = false;
        if (!configured) {
            GTE_SET_LDS((gemm_p3_nt_ring_kernel<2, 4, 2, 2, 3, 1, 2>), shm);
            configured = true;
        }
        hipLaunchKernelGGL((gemm_p3_nt_ring_kernel<2, 4, 2, 2, 3, 1, 2>), dim3((unsigned)nb), dim3(512), shm, s, p);
    }
    int rc = gte::check_launch("gemm_p3_nt_smallk_bwd");
    if (rc != GTE_OK) return rc;
    if (gte::defer_fold(p.sk_part_dw, n * K, nb, (int)n, (int)K, dW, lddw)) {
        gte::defer_fold(p.ln_part, 3 * n, nb, 1, (int)n, dgamma, n);
        gte::defer_fold(p.ln_part + n, 3 * n, nb, 1, (int)n, dbeta, n);
        gte::defer_fold(p.ln_part + 2 * n, 3 * n, nb, 1, (int)n, dbias, n);
        return GTE_OK;
    }
    // (no deferral open: fold here; dW is written packed, lddw == K)
    if (lddw != K) return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_nt_smallk_bwd: outside a fold deferral dW must be packed (lddw == k1 + k2)");
    hipLaunchKernelGGL(p3_colsum_fold_kernel, dim3((unsigned)gte::ceil_div(n * K, 256)), dim3(256), 0, s, p.sk_part_dw, (long long)n * K, nb,
                       (int)(n * K), dW);
    float* outs[3] = {dgamma, dbeta, dbias};
    for (int i = 0; i < 3; ++i)
        if (outs[i])
            hipLaunchKernelGGL(p3_colsum_fold_kernel, dim3((unsigned)gte::ceil_div(n, 256)), dim3(256), 0, s, p.ln_part + i * n,
                               (long long)3 * n, nb, (int)n, outs[i]);
    return gte::check_launch("gemm_p3_nt_smallk_bwd fold");
}

extern "C" int64_t gte_gemm_p3_tn_workspace_bytes(int64_t m, int64_t n, int64_t nseg, int64_t k) {
    if (m <= 0 || n <= 0) return 256;
    const TnPlan pl = tn_plan(m, n, nseg, k);
    return pl.splits_bound > 1 ? gte::round_up((int64_t)pl.splits_bound * m * n * 4, 256) : 256;
}

// C[m, n] = a^T b over k rows: a = P3 [k][m], b = P3 [k][n].  nseg > 0: two column segments of nseg columns (n == 2 nseg),
// C[:, 0:nseg] = a^T b, C[:, nseg:] = a2^T b2 (a2 / b2 nullable: the operand of segment 0).  The split-K fold joins an open
// fold deferral (gte_fold_defer_begin), else it runs as its own launch.
extern "C" int gte_gemm_p3_tn(const void* a, int64_t lda, const void* a2, int64_t lda2, const void* b, int64_t ldb, const void* b2,
                              int64_t ldb2, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n, int64_t k, void* workspace,
                              int64_t workspace_bytes, void* stream);

namespace {
__global__ void __launch_bounds__(256)
p3_fold_kernel(const float* __restrict__ slab, int splits, long long mn, int N, float* __restrict__ C, long long ldc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mn) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(long long)k * mn + i];
    const long long r = i / N;
    C[r * ldc + (i - r * N)] = s;
}
}  // namespace

static int gemm_p3_tn_impl(const void* a, int64_t lda, const void* a2, int64_t lda2, const void* b, int64_t ldb, const void* b2,
                           int64_t ldb2, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n, int64_t k, void* workspace,
                           int64_t workspace_bytes, void* stream, const int32_t* b_rows, int64_t n_res_rows) {
    if (m <= 0 || n <= 0 || k < 0 || nseg < 0 || m > INT32_MAX || n > INT32_MAX || k > INT32_MAX || (nseg > 0 && n != 2 * nseg))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_tn: bad sizes");
    if (!a || !b || !c) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_tn: null pointer");
    const int64_t bcols = nseg > 0 ? nseg : n;
    if (lda < p3::row_bytes(m) || ldb < p3::row_bytes(bcols) || (a2 && lda2 < p3::row_bytes(m)) || (b2 && ldb2 < p3::row_bytes(bcols)) ||
        ldc < n)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_tn: leading dimension too small");
    if (lda >= (1 << 23) || ldb >= (1 << 23) || lda2 >= (1 << 23) || ldb2 >= (1 << 23) || (m + 128) * n * 4 >= ((int64_t)1 << 31) ||
        (m + 128) * ldc * 4 >= ((int64_t)1 << 31))
        return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_tn: row strides must be < 8 MB and the output < 2 GB");
    hipStream_t s = gte::as_stream(stream);
    if (k == 0) {
        for (int64_t r = 0; r < m; ++r)
            if (hipMemsetAsync(c + r * ldc, 0, (size_t)n * 4, s) != hipSuccess) return gte::fail(GTE_ERR_LAUNCH, "gemm_p3_tn: memset failed");
        return GTE_OK;
    }
    const TnPlan pl = tn_plan(m, n, nseg, k);
    P3Gemm p = {};
    p.A1 = (const char*)a; p.lda1 = lda; p.An2 = (const char*)a2; p.ldan2 = lda2;
    p.B = (const char*)b; p.ldb = ldb; p.Bn2 = (const char*)b2; p.ldbn2 = ldb2;
    p.Nseg = (int)nseg; p.M = (int)m; p.N = (int)n; p.K = (int)k; p.C = c; p.ldc = ldc;
    p.splits = pl.splits; p.stages_per_split = pl.stages_per_split;
    if (b_rows) {
        if (b2 && ldb2 != ldb) return gte::fail(GTE_ERR_UNSUPPORTED, "gemm_p3_tn_rows: both resident images need the same row stride");
        if (n_res_rows <= 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_tn_rows: empty resident image");
        p.rowsB = b_rows; p.res_bytes = n_res_rows * ldb;
        p.rows64 = rows64_needed((n_res_rows + 1) * ldb);
    }
    if (pl.splits > 1) {
        const int64_t need = (int64_t)pl.splits * m * n * 4;
        if (!workspace || workspace_bytes < need)
            return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "gemm_p3_tn: split-K needs %lld workspace bytes, got %lld", (long long)need,
                             (long long)workspace_bytes);
        p.slab = reinterpret_cast<float*>(workspace);
    }
    const int64_t tiles = gte::ceil_div(m, 128) * (nseg > 0 ? 2 * gte::ceil_div(nseg, 128) : gte::ceil_div(n, 128));
    constexpr int shm_small = 3 * 6 * 4 * 1024;
    static bool configured = false;
    if (!configured) {
        GTE_SET_LDS((gemm_p3_tn_kernel<2, 2, 2>), shm_small);
        GTE_SET_LDS((gemm_p3_tn_kernel<2, 2, 2, true>), shm_small);
        GTE_SET_LDS((gemm_p3_tn_kernel<2, 2, 2, true, true>), shm_small);
        configured = true;
    }
    const dim3 grid((unsigned)(tiles * pl.splits));
    if (p.rowsB && p.rows64) hipLaunchKernelGGL((gemm_p3_tn_kernel<2, 2, 2, true, true>), grid, dim3(256), shm_small, s, p);
    else if (p.rowsB) hipLaunchKernelGGL((gemm_p3_tn_kernel<2, 2, 2, true>), grid, dim3(256), shm_small, s, p);
    else hipLaunchKernelGGL((gemm_p3_tn_kernel<2, 2, 2>), grid, dim3(256), shm_small, s, p);
    int rc = gte::check_launch("gemm_p3_tn");
    if (rc != GTE_OK || pl.splits <= 1) return rc;
    const int64_t mn = m * n;
    if (gte::defer_fold(p.slab, mn, pl.splits, (int)m, (int)n, c, ldc)) return GTE_OK;
    hipLaunchKernelGGL(p3_fold_kernel, dim3((unsigned)gte::ceil_div(mn, 256)), dim3(256), 0, s, p.slab, pl.splits, (long long)mn, (int)n,
                       c, (long long)ldc);
    return gte::check_launch("gemm_p3_tn fold");
}

extern "C" int gte_gemm_p3_tn(const void* a, int64_t lda, const void* a2, int64_t lda2, const void* b, int64_t ldb, const void* b2,
                              int64_t ldb2, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n, int64_t k, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    return gemm_p3_tn_impl(a, lda, a2, lda2, b, ldb, b2, ldb2, nseg, c, ldc, m, n, k, workspace, workspace_bytes, stream, nullptr, 0);
}

// ... with b = the rows b_rows[0 .. k) of a RESIDENT P3 image b_res [n_res_rows][.] (the input layer's dW straight from the
// resident features).  b_rows must hold k rounded up to 16, plus 1, entries; the entries past k = n_res_rows (a row past the
// image: zeros).  Both column segments read the same image.  An image of 4 GB or more is read through 64-bit addresses without a
// range check: row n_res_rows must then exist in the allocation and hold zeros.
extern "C" int gte_gemm_p3_tn_rows(const void* a, int64_t ldpa, const void* a2, int64_t ldpa2, const void* b_res, int64_t ldpb,
                                   const int32_t* b_rows, int64_t n_res_rows, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n,
                                   int64_t k, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!b_rows) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_tn_rows: null row map");
    return gemm_p3_tn_impl(a, ldpa, a2, ldpa2, b_res, ldpb, nullptr, 0, nseg, c, ldc, m, n, k, workspace, workspace_bytes, stream, b_rows,
                           n_res_rows);
}

// ... with TWO resident images behind the map: C[:, 0:nseg] = a^T b_res[rows], C[:, nseg:] = a^T b2_res[rows] (the input features
// and their cached mean aggregate: dW = [dz^T x | dz^T ahn] of an aggregate-first input layer, no q aggregation).  Same row stride.
extern "C" int gte_gemm_p3_tn_rows2(const void* a, int64_t ldpa, const void* b_res, int64_t ldpb, const void* b2_res, int64_t ldpb2,
                                    const int32_t* b_rows, int64_t n_res_rows, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n,
                                    int64_t k, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!b_rows || !b2_res || nseg <= 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_p3_tn_rows2: null row map / second image, or nseg <= 0");
    return gemm_p3_tn_impl(a, ldpa, nullptr, 0, b_res, ldpb, b2_res, ldpb2, nseg, c, ldc, m, n, k, workspace, workspace_bytes, stream, b_rows,
                           n_res_rows);
}
