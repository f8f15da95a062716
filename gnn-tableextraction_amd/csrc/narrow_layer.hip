// The class-count-wide output layer of GcnSAGE (hidden -> n_classes = 9, no LayerNorm, no activation:
// src/components/graphs/models.py:101-103) in transform-then-aggregate form.
//
// The reference computes  logits = [h | norm * (A_w h)] W^T + b  (models.py:53-72).  With W = [W_s | W_n]:
//     logits = h W_s^T + b + norm * A_w (h W_n^T)                        (linearity of the aggregation)
// so the aggregation runs on C = 9 columns instead of F = 256, and in the backward
//     q  = A_w^T (norm * dlogits)                                       [N, C]   (9-wide transpose aggregation)
//     dW = [dlogits^T h | q^T h],   dh = dlogits W_s + q W_n,   dbias = colsum(dlogits)
// every wide operand is h itself: the forward reads h once, the backward reads h once and writes dh once.
// The generic path needed 11 launches here (aggregate 256-wide, MFMA GEMM with N = 9 padded to 32, two skinny
// dW GEMMs + reductions, two dX GEMMs, a 256-wide transpose aggregation) -- ~174 us per step at 24.5 k nodes
// against ~45 us for this path (forward 14, backward 22-25, 9-wide aggregations 2 x 6).  HBM-bound work: 25 MB in for
// the forward, 25 MB in + 25 MB out for the backward at F = 256.
//
// Kernels in this file:
//   narrow_fwd_kernel / narrow_bwd_kernel<NJ, NCT>      plain FMA + wave reductions, any F <= 256   (fallback, F % 8 != 0)
//   narrow_fwd_mfma_kernel / narrow_bwd_mfma_kernel     the same products on the matrix pipe        (F % 8 == 0: the shipped path)
//   head_agg_ce_kernel                                  logits = t_self + aggregate(t_neigh), weighted CE, unnormalised gradient
//   narrow_bwd_mfma_kernel<NCT, true>                   ... plus the LayerNorm backward of the layer below (built, off by default)
// Plain-kernel mapping: one wave64 per node row, lane l owns features l, l+64, ... (NJ <= 4, F <= 256); the 2C weight
// rows live in registers (2C*NJ <= 128 VGPRs), row dot products are wave reductions, the dW partial sums accumulate
// in registers over the rows a wave owns and are folded through LDS per block, then across blocks in a fixed
// order (no atomics: deterministic).
#include "gte_common.h"
#include "ce_fold.h"
#include "p3.h"

#include <stdlib.h>

namespace {

constexpr int NC_MAX = 16;          // max output width of the narrow path
constexpr int NB_MAX = 512;         // blocks of the backward kernel (partials folded afterwards)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// t_self[r, c] = <h[r], W[c, 0:F]> + bias[c],  t_neigh[r, c] = <h[r], W[c, F:2F]>
template <int NJ, int NCT>          // NCT: compile-time bound on C (register arrays are sized by it)
__global__ void __launch_bounds__(256)
narrow_fwd_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ W, int64_t ldw,
                  const float* __restrict__ bias, float* __restrict__ t_self, int64_t lds_, float* __restrict__ t_neigh,
                  int64_t ldn, int n, int F, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ws[NCT][NJ], wn[NCT][NJ];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = lane + 64 * j;
            const bool ok = c < C && k < F;
            ws[c][j] = ok ? W[(int64_t)c * ldw + k] : 0.f;
            wn[c][j] = ok ? W[(int64_t)c * ldw + F + k] : 0.f;
        }
    for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
        float hv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int k = lane + 64 * j; hv[j] = k < F ? h[(int64_t)r * ldh + k] : 0.f; }
        float outs = 0.f, outn = 0.f;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            if (c < C) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { a = fmaf(hv[j], ws[c][j], a); b = fmaf(hv[j], wn[c][j], b); }
                a = wave_sum(a);
                b = wave_sum(b);
                if (lane == c) { outs = a; outn = b; }
            }
        }
        if (lane < C) {
            t_self[(int64_t)r * lds_ + lane] = outs + (bias ? bias[lane] : 0.f);
            t_neigh[(int64_t)r * ldn + lane] = outn;
        }
    }
}

// dh[r, k] = sum_c dl[r,c] W[c,k] + q[r,c] W[c,F+k];  partial dW[c, k] += dl[r,c] h[r,k], dW[c, F+k] += q[r,c] h[r,k]
template <int NJ, int NCT>
__global__ void __launch_bounds__(256)
narrow_bwd_kernel(const float* __restrict__ dl, int64_t lddl, const float* __restrict__ q, int64_t ldq,
                  const float* __restrict__ h, int64_t ldh, const float* __restrict__ W, int64_t ldw,
                  float* __restrict__ dh, int64_t lddh, float* __restrict__ partial, int n, int F, int C) {
    extern __shared__ float red[];                     // [2C][64*NJ] + [C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int FW = 64 * NJ;
    float ws[NCT][NJ], wn[NCT][NJ], gs[NCT][NJ], gn[NCT][NJ];
    float gb = 0.f;                                    // lane c accumulates dbias[c]
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = lane + 64 * j;
            const bool ok = c < C && k < F;
            ws[c][j] = ok ? W[(int64_t)c * ldw + k] : 0.f;
            wn[c][j] = ok ? W[(int64_t)c * ldw + F + k] : 0.f;
            gs[c][j] = 0.f; gn[c][j] = 0.f;
        }
    for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
        float hv[NJ], o[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int k = lane + 64 * j; hv[j] = k < F ? h[(int64_t)r * ldh + k] : 0.f; o[j] = 0.f; }
        const float my_dl = lane < C ? dl[(int64_t)r * lddl + lane] : 0.f;
        const float my_q = lane < C ? q[(int64_t)r * ldq + lane] : 0.f;
        gb += my_dl;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            if (c < C) {
                const float d = __shfl(my_dl, c, 64), qq = __shfl(my_q, c, 64);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    o[j] = fmaf(d, ws[c][j], o[j]);
                    o[j] = fmaf(qq, wn[c][j], o[j]);
                    gs[c][j] = fmaf(d, hv[j], gs[c][j]);
                    gn[c][j] = fmaf(qq, hv[j], gn[c][j]);
                }
            }
        }
        if (dh) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) { const int k = lane + 64 * j; if (k < F) dh[(int64_t)r * lddh + k] = o[j]; }
        }
    }
    // fold the four waves through LDS (wave 0 writes, the others add in wave order), then write the block partial
    float* bsum = red + 2 * C * FW;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
                if (c < C) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int k = lane + 64 * j;
                        float* ps = red + (size_t)c * FW + k;
                        float* pn = red + (size_t)(C + c) * FW + k;
                        if (w == 0) { *ps = gs[c][j]; *pn = gn[c][j]; } else { *ps += gs[c][j]; *pn += gn[c][j]; }
                    }
                }
            if (lane < C) { if (w == 0) bsum[lane] = gb; else bsum[lane] += gb; }
        }
        __syncthreads();
    }
    float* pp = partial + (int64_t)blockIdx.x * (2 * C * F + C);
    for (int i = threadIdx.x; i < 2 * C * F; i += 256) {
        const int c2 = i / F, k = i - c2 * F;
        pp[i] = red[(size_t)c2 * FW + k];
    }
    if (threadIdx.x < C) pp[2 * C * F + threadIdx.x] = bsum[threadIdx.x];
}

// dW[c, seg*F + k] = sum_b partial[b][(seg*C + c)*F + k];  dbias[c] = sum_b partial[b][2CF + c]   (fixed order)
__global__ void __launch_bounds__(256)
narrow_fold_kernel(const float* __restrict__ partial, int nblk, int F, int C, float* __restrict__ dW, int64_t lddw,
                   float* __restrict__ dbias) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int total = 2 * C * F + C;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < total) {
#pragma unroll 4
        for (int b = sl; b < nblk; b += 4) s += partial[(int64_t)b * total + i];
    }
    part[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
        const float v = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (i < 2 * C * F) {
            const int c2 = i / F, k = i - c2 * F;
            const int seg = c2 / C, c = c2 - seg * C;
            dW[(int64_t)c * lddw + seg * F + k] = v;
        } else if (dbias) {
            dbias[i - 2 * C * F] = v;
        }
    }
}

// ------------------------------- MFMA versions (F % 8 == 0) -----------------------------------------
// The plain kernels above spend their time in wave reductions (18 per row) and scalar FMA chains: 37 us
// (forward) and 41 us (backward) at 24.5 k nodes, F = 256, against ~6 / ~13 us of HBM time.  The same
// products on the matrix pipe (v_mfma_f32_32x32x2_f32, exact fp32):
//   forward : [32 rows] x [32 cols = 16 self | 16 neigh]  per wave and row block, K = F; h fragments straight from
//             global memory (16-byte loads, lane = (row, k-half)), W fragments from an LDS image;
//   backward: dh  = DLQ[32 rows][2 NCT] x Wst[2 NCT][F]            (K = 2 NCT: dl columns, then q columns)
//             dW += DLQ^T[2 NCT][32 rows] x h[32 rows][F]          (K = the 32 rows of the block; accumulators
//             persist over the row blocks a wave owns, then fold wave -> block -> grid in a fixed order).
//   Column permutation: lane i of tile j holds column 4 i + j (+128 for tiles 4..7), so one 16-byte load of h
//   (or store of dh) serves four tiles and a half-wave touches 512 contiguous bytes of one row.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4n { float x, y, z, w; };

// stage rows of W into an LDS image: image row m <- W[src_row(m)][seg(m) * F + 0 .. F), zero when src_row >= C
// HALF: image rows [0, HALF) come from W_s (seg 0), rows [HALF, 2 HALF) from W_n (seg 1)
// Ft: the TRUE feature width of W ([C][2 Ft]: W_s = columns [0, Ft), W_n = [Ft, 2 Ft)); F >= Ft is the width the kernel computes
// over (rows of h padded with zeros up to it -- hidden widths that are not a multiple of 8: round 5); image columns Ft .. F-1 zero
template <int ROWS, int HALF>
__device__ __forceinline__ void stage_w_image(float* __restrict__ img, int FP, const float* __restrict__ W, int64_t ldw, int F,
                                              int C, int Ft) {
    const int f4 = F / 4;
    for (int idx = threadIdx.x; idx < ROWS * f4; idx += blockDim.x) {
        const int m = idx / f4, k = (idx - m * f4) * 4;
        const int seg = m >= HALF ? 1 : 0, c = m - seg * HALF;
        float* dst = img + m * FP + k;
        if (c < C && k + 3 < Ft) {
            const float* src = W + (int64_t)c * ldw + seg * Ft + k;
            const float x = src[0], y = src[1], z = src[2], w = src[3];
            dst[0] = x; dst[1] = y; dst[2] = z; dst[3] = w;
        } else if (c < C && k < Ft) {
            const float* src = W + (int64_t)c * ldw + seg * Ft + k;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = k + e < Ft ? src[e] : 0.f;
        } else {
            dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f; dst[3] = 0.f;
        }
    }
}

// LNF: the input is the PRE-LayerNorm z of the layer below; the kernel normalises its rows in registers (statistics and
// affine arithmetic of ln_relu_fwd_vec_kernel; a row's two halves sit in lanes i and i + 32), writes y = relu?(LN(z)) and the
// row statistics for the backward, and feeds the products from the normalised registers: the layer below needs no
// LayerNorm launch (gte_sage_narrow_fwd_ln).
struct LnForward { const float* gamma; const float* beta; float eps; int relu; float* y; int64_t ldy; float* stats; };

template <bool LNF>
__global__ void __launch_bounds__(256)
narrow_fwd_mfma_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ W, int64_t ldw,
                       const float* __restrict__ bias, float* __restrict__ t_self, int64_t lds_,
                       float* __restrict__ t_neigh, int64_t ldn, int n, int F, int C, const LnForward lnf, int Ft) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];     // [32][F + 4]: col c < 16 -> W_s row c, 16 + c -> W_n row c
    const int FP = F + 4;                                          // LNF: + gamma[F], beta[F] behind the image
    stage_w_image<32, 16>(Wl, FP, W, ldw, F, C, Ft);
    float* GB = Wl + 32 * FP;
    if constexpr (LNF) {
        for (int k = threadIdx.x; k < F; k += blockDim.x) { GB[k] = lnf.gamma[k]; GB[F + k] = lnf.beta[k]; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, hh = lane >> 5;
    const int nblk = (n + 31) / 32, kbs = F / 8;
    const float* wp = Wl + i * FP + hh * 4;
    const int cv = i & 15, seg = i >> 4;
    const float bv = (seg == 0 && bias && cv < C) ? bias[cv] : 0.f;
    const int wpb = blockDim.x >> 6;                      // waves per workgroup (launch parameter)
    for (int rb = blockIdx.x * wpb + wave; rb < nblk; rb += gridDim.x * wpb) {
        const int row = min(rb * 32 + i, n - 1);
        const float* hp = h + (int64_t)row * ldh + hh * 4;
        // all of the row's 16-byte loads are issued before the first MFMA (one wave per SIMD at 24 k rows: nothing else
        // hides the latency); two accumulator chains (even / odd k-blocks) so consecutive MFMAs do not depend on each other
        f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
        f4n a[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) a[u] = *reinterpret_cast<const f4n*>(hp + min(u, kbs - 1) * 8);
        if constexpr (LNF) {
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u) if (u < kbs) sm += (a[u].x + a[u].y) + (a[u].z + a[u].w);
            sm += __shfl_xor(sm, 32, 64);
            const float mean = sm * (1.0f / (float)F);
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (u < kbs) {
                    const float d0 = a[u].x - mean, d1 = a[u].y - mean, d2 = a[u].z - mean, d3 = a[u].w - mean;
                    q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
                }
            q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q * (1.0f / (float)F) + lnf.eps);
            const bool rowok = rb * 32 + i < n;
            if (rowok && hh == 0 && lnf.stats) { lnf.stats[row] = mean; lnf.stats[n + row] = rstd; }
            float* yp = lnf.y + (int64_t)row * lnf.ldy + hh * 4;
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (u < kbs) {
                    const f32x4n g = *reinterpret_cast<const f32x4n*>(GB + u * 8 + hh * 4);
                    const f32x4n b = *reinterpret_cast<const f32x4n*>(GB + F + u * 8 + hh * 4);
                    f4n o;
                    o.x = fmaf((a[u].x - mean) * rstd, g.x, b.x); o.y = fmaf((a[u].y - mean) * rstd, g.y, b.y);
                    o.z = fmaf((a[u].z - mean) * rstd, g.z, b.z); o.w = fmaf((a[u].w - mean) * rstd, g.w, b.w);
                    if (lnf.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    a[u] = o;
                    if (rowok) *reinterpret_cast<f4n*>(yp + u * 8) = o;
                }
        }
#pragma unroll
        for (int u = 0; u < 32; u += 2) {
            if (u < kbs) {
                const f32x4n b = *reinterpret_cast<const f32x4n*>(wp + u * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b.w, acc, 0, 0, 0);
            }
            if (u + 1 < kbs) {
                const f32x4n b = *reinterpret_cast<const f32x4n*>(wp + (u + 1) * 8);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1].x, b.x, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1].y, b.y, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1].z, b.z, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1].w, b.w, acc2, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
        if (cv < C) {
            float* outp = seg ? t_neigh : t_self;
            const int64_t ldo = seg ? ldn : lds_;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (rr < n) outp[(int64_t)rr * ldo + cv] = acc[r] + bv;
            }
        }
    }
}

// The same forward on v_mfma_f32_16x16x4_f32 with 16-row blocks, one per wave, 8 waves per workgroup (F % 16 == 0): twice
// the waves of the 32-row version for the same matrix-pipe time per row (32 cycles per 16x16x4 against 64 per 32x32x2), so
// a SIMD holds two waves and one wave's row loads / y stores run under the other's MFMAs -- the 32-row version is a single
// serial chain per SIMD (load 32 KB, normalise, 128 MFMAs, store): 22 us at 24.5 k rows against ~10 us of HBM time.
//   lane = (row i = lane & 15, k-quarter kq = lane >> 4): a[u] = row i, k = 16 u + 4 kq .. + 3 (one 16-byte load);
//   MFMA t of k-block u multiplies k = 16 u + 4 kq + t: A and B agree on that order.  C/D: col = lane & 15,
//   row = 4 (lane >> 4) + r.  Two column tiles: W_s rows (image rows 0..15) and W_n rows (16..31).
typedef float f32x4acc __attribute__((ext_vector_type(4)));

template <bool LNF>
__global__ void __launch_bounds__(512)
narrow_fwd_mfma16_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ W, int64_t ldw,
                         const float* __restrict__ bias, float* __restrict__ t_self, int64_t lds_,
                         float* __restrict__ t_neigh, int64_t ldn, int n, int F, int C, const LnForward lnf, int Ft) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];
    const int FP = F + 4;
    stage_w_image<32, 16>(Wl, FP, W, ldw, F, C, Ft);
    float* GB = Wl + 32 * FP;
    if constexpr (LNF) {
        for (int k = threadIdx.x; k < F; k += blockDim.x) { GB[k] = lnf.gamma[k]; GB[F + k] = lnf.beta[k]; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
    const int nblk = (n + 15) / 16, kbs = F / 16;
    const float* wps = Wl + i * FP + kq * 4;
    const float* wpn = Wl + (16 + i) * FP + kq * 4;
    const float bv = (bias && i < C) ? bias[i] : 0.f;
    const int wpb = blockDim.x >> 6;
    for (int rb = blockIdx.x * wpb + wave; rb < nblk; rb += gridDim.x * wpb) {
        const int row = min(rb * 16 + i, n - 1);
        const float* hp = h + (int64_t)row * ldh + kq * 4;
        f4n a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = *reinterpret_cast<const f4n*>(hp + min(u, kbs - 1) * 16);
        if constexpr (LNF) {
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) if (u < kbs) sm += (a[u].x + a[u].y) + (a[u].z + a[u].w);
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            const float mean = sm * (1.0f / (float)F);
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < kbs) {
                    const float d0 = a[u].x - mean, d1 = a[u].y - mean, d2 = a[u].z - mean, d3 = a[u].w - mean;
                    q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
                }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q * (1.0f / (float)F) + lnf.eps);
            const bool rowok = rb * 16 + i < n;
            if (rowok && kq == 0 && lnf.stats) { lnf.stats[row] = mean; lnf.stats[n + row] = rstd; }
            float* yp = lnf.y + (int64_t)row * lnf.ldy + kq * 4;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < kbs) {
                    const f32x4n g = *reinterpret_cast<const f32x4n*>(GB + u * 16 + kq * 4);
                    const f32x4n b = *reinterpret_cast<const f32x4n*>(GB + F + u * 16 + kq * 4);
                    f4n o;
                    o.x = fmaf((a[u].x - mean) * rstd, g.x, b.x); o.y = fmaf((a[u].y - mean) * rstd, g.y, b.y);
                    o.z = fmaf((a[u].z - mean) * rstd, g.z, b.z); o.w = fmaf((a[u].w - mean) * rstd, g.w, b.w);
                    if (lnf.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    a[u] = o;
                    if (rowok) *reinterpret_cast<f4n*>(yp + u * 16) = o;
                }
        }
        f32x4acc as = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u < kbs) {
                const f32x4n bs = *reinterpret_cast<const f32x4n*>(wps + u * 16);
                const f32x4n bn = *reinterpret_cast<const f32x4n*>(wpn + u * 16);
                as = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, bs.x, as, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, bn.x, an, 0, 0, 0);
                as = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, bs.y, as, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, bn.y, an, 0, 0, 0);
                as = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, bs.z, as, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, bn.z, an, 0, 0, 0);
                as = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, bs.w, as, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, bn.w, an, 0, 0, 0);
            }
        }
        if (i < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = rb * 16 + 4 * kq + r;
                if (rr < n) {
                    t_self[(int64_t)rr * lds_ + i] = as[r] + bv;
                    t_neigh[(int64_t)rr * ldn + i] = an[r];
                }
            }
        }
    }
}

// Backward.  A workgroup walks 32-row blocks; its four waves split the F columns (wave w owns the 64-column slice
// starting at 64 w, lane i of tile j holds column 64 w + 2 i + j), so a wave keeps 2 dW tiles + 2 dh tiles in
// accumulators (~120 registers: 4 waves per SIMD hide the global latency) and the dW partials of a workgroup need no
// fold across its waves.  NCT: compile-time bound on C (4, 8, 12, 16).
// Optional LayerNorm(+ReLU) backward of the layer BELOW, fused behind the dh product (LNB kernels): h = relu(LN(z)) was
// that layer's output; instead of dh the kernel writes dz = LN'(z) . (relu mask . dh) and leaves the column partials
// {sum g xhat, sum g, sum dz} per workgroup in lnpart[block][3][F] -- dh (25 MB at 24 k x 256) is never written or read.
struct LnBackward {
    const float* z; int64_t ldz; const float* stats; const float* gamma; const float* beta; int relu; float* lnpart;
    char* dzp3; int64_t ldp3;                   // dz additionally as a P3 image (csrc/p3.h)
};

// LNB: 0 plain; 2 the dh tile goes through LDS and every wave takes whole rows of it in the layout and with the arithmetic of
// ln_relu_bwd_vec_kernel (16-byte accesses, dz as fp32 + P3 image: bit for bit gte_sage_narrow_bwd + gte_ln_relu_bwd_p3).
// (Round 2's form 1 -- the LayerNorm backward in the accumulator layout, row sums across the four waves -- measured slower than
// two launches, 41.6 against 39 us at 24 k x 256, and is gone; so is form 2's option of gathering q = A_w^T (norm dl) from the
// out-edge CSR inside the kernel: 18 us of dependent loads for the 6 us launch it saved.  profiles/r03/gemm_p3.md keeps the numbers.)
// MSK (with Ft < F): the rows are PADDED -- Ft true columns, zeros up to F (a multiple of 16): W is [C][2 Ft], the dW / LayerNorm
// partials are written compact ([.][Ft]), the LayerNorm backward runs over the Ft true columns with per-element validity (the
// arithmetic of ln_relu_bwd_gen_kernel) and writes zeros into the padding of dz and of its image
#ifndef NB_ABL
#define NB_ABL 0          // measurement builds (profiles/debug/build_variant.sh): 1 no dW products, 2 no LayerNorm phase, 4 no dh products,
#endif                    // 8 no dz stores, 16 no z loads, 32 no CE fold / alpha
template <int NCT, int LNB = 0, bool MSK = false>
__global__ void __launch_bounds__(256, LNB == 2 ? 2 : 1)
narrow_bwd_mfma_kernel(const float* __restrict__ dl, int64_t lddl, const float* __restrict__ q, int64_t ldq,
                       const float* __restrict__ h, int64_t ldh, const float* __restrict__ W, int64_t ldw,
                       float* __restrict__ dh, int64_t lddh, float* __restrict__ partial, int n, int F, int C,
                       const float* __restrict__ ce_partial, int64_t ce_blocks, float grad_scale, float* __restrict__ out3,
                       const LnBackward lnb = LnBackward{}, int Ft_ = 0) {
    const int Ft = MSK ? Ft_ : F;
    constexpr int KD = 2 * NCT;                 // compact K of the dh product / M of the dW product
    constexpr int DP = KD + 1;                  // row stride of the DLQ image (odd: conflict-free read both ways)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int FP = F + 4;
    float* Wst = sm;                            // [KD][FP]: row kk < NCT -> W_s row kk, NCT + c -> W_n row c
    float* D = sm + KD * FP;                    // [32][DP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, hh = lane >> 5;
    const int nblk = (n + 31) / 32;
    const int col = 64 * wave + 2 * i;          // this lane's two columns: col, col + 1
    const bool colok = col < F;                 // F % 8 == 0: both or neither

    f32x16 gw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) gw[j][r] = 0.f;
    float gb = 0.f;                             // wave 0, lane c (< C): dbias[c]
    struct __attribute__((packed, aligned(4))) f2n { float x, y; };
    // LNB == 2: lane l owns columns 4 l .. 4 l + 3 of the rows its wave takes
    const int j4 = 4 * lane;
    const bool okc = MSK ? j4 < ((Ft + 3) & ~3) : j4 < F;      // the lane holds a valid column
    const bool imc = MSK ? j4 < F : okc;                        // ... or an image column of the padding (written as zero)
    bool oke[4];
    float gam4[4] = {1.f, 1.f, 1.f, 1.f}, bet4[4] = {0.f, 0.f, 0.f, 0.f};
    float c_dg[4] = {0.f, 0.f, 0.f, 0.f}, c_db[4] = {0.f, 0.f, 0.f, 0.f}, c_dz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) oke[e] = MSK ? j4 + e < Ft : okc;
    if constexpr (LNB == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (oke[e]) { gam4[e] = lnb.gamma[j4 + e]; bet4[e] = lnb.beta[j4 + e]; }
    }

    // Row-block pipeline: the h rows (for the dW product) and the dl / q values of the NEXT row block are requested before
    // the current block is computed; the first block's requests go out before the CE fold and the W image are done.
    constexpr int ND = (32 * KD + 255) / 256;
    // LNB == 2 runs two workgroups per CU (256 registers per lane): the h rows of a block are requested at its top -- their only use
    // is the dW product at its end, a whole block later -- instead of a block ahead (32 registers), and the other workgroup of the CU
    // covers what latency is left.  Measured before (one workgroup per CU, 316 registers): the phases of a block simply add up --
    // skeleton 11 + dh 4 + LayerNorm 18 + dW 8 us of 38 (profiles/r05/narrow_bwd_abl.txt).
    constexpr bool AHEAD = LNB != 2;
    f2n hvn[AHEAD ? 16 : 1];
    float dvn[ND];
    auto request = [&](int rbn) {
        const int r0 = rbn * 32;
        if constexpr (AHEAD) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                hvn[s] = f2n{0.f, 0.f};
                if (colok) hvn[s] = *reinterpret_cast<const f2n*>(h + (int64_t)min(r0 + 2 * s + hh, n - 1) * ldh + col);
            }
        }
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int idx = threadIdx.x + 256 * t;
            const int r = idx / KD, kk = idx - r * KD, c = kk < NCT ? kk : kk - NCT;
            dvn[t] = 0.f;
            if (idx < 32 * KD && r0 + r < n && c < C) {
                if (kk < NCT) dvn[t] = dl[(int64_t)(r0 + r) * lddl + c];
                else dvn[t] = q[(int64_t)(r0 + r) * ldq + c];
            }
        }
    };
    if ((int)blockIdx.x < nblk) request(blockIdx.x);
    // fused head (gte_head_agg_ce): dl / q arrive WITHOUT the 1 / sum(w) of the weighted cross-entropy; every block folds
    // the CE partials itself (fixed order, same value everywhere) and scales its DLQ image; block 0 publishes out3
    float alpha = 1.f;
    if (ce_partial && !(NB_ABL & 32)) {
        __shared__ double ce_red[3][gte_ce::kCeBlock];
        gte_ce::ce_fold(ce_partial, ce_blocks, ce_red);
        const float wsum = (float)ce_red[1][0];
        alpha = wsum > 0.f ? grad_scale / wsum : 0.f;
        if (blockIdx.x == 0 && threadIdx.x == 0 && out3) gte_ce::ce_write_out3(ce_red, out3);
    }
    stage_w_image<KD, NCT>(Wst, FP, W, ldw, F, C, Ft);

    for (int rb = blockIdx.x; rb < nblk; rb += gridDim.x) {
        const int row0 = rb * 32;
        f2n hv[16];
        float dv[ND];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if constexpr (AHEAD) hv[s] = hvn[s];
            else {
                hv[s] = f2n{0.f, 0.f};
                if (colok) hv[s] = *reinterpret_cast<const f2n*>(h + (int64_t)min(row0 + 2 * s + hh, n - 1) * ldh + col);
            }
        }
#pragma unroll
        for (int t = 0; t < ND; ++t) dv[t] = dvn[t];
        if (rb + (int)gridDim.x < nblk) request(rb + gridDim.x);
        // LNB == 2: the z rows (and statistics) of the eight rows this wave takes of the block: rows wave, wave + 4, ...
        float zr[LNB == 2 ? 8 : 1][4], mu2[LNB == 2 ? 8 : 1], rs2[LNB == 2 ? 8 : 1];
        if constexpr (LNB == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = min(row0 + wave + 4 * u, n - 1);
                struct __attribute__((packed, aligned(4))) f4n { float x, y, z, w; };
                f4n t{0.f, 0.f, 0.f, 0.f};
                if (okc && !(NB_ABL & 16)) t = *reinterpret_cast<const f4n*>(lnb.z + (int64_t)rr * lnb.ldz + j4);
                zr[u][0] = t.x; zr[u][1] = t.y; zr[u][2] = t.z; zr[u][3] = t.w;
                mu2[u] = lnb.stats[rr]; rs2[u] = lnb.stats[n + rr];
            }
        }
        __syncthreads();                        // the previous block's D reads are done (and W image staged)
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int idx = threadIdx.x + 256 * t;
            const int r = idx / KD, kk = idx - r * KD;
            if (idx < 32 * KD) D[r * DP + kk] = dv[t] * alpha;
        }
        __syncthreads();
        if (wave == 0 && i < C && hh == 0) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) gb += D[r * DP + i];
        }
        if (dh) {
            f32x16 o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
#pragma unroll
            for (int s = 0; s < ((NB_ABL & 4) ? 1 : NCT); ++s) {
                const float a = D[i * DP + 2 * s + hh];
                float b0 = 0.f, b1 = 0.f;
                if (colok) { const float2 b = *reinterpret_cast<const float2*>(Wst + (2 * s + hh) * FP + col); b0 = b.x; b1 = b.y; }
                o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, o[1], 0, 0, 0);
            }
            if constexpr (LNB == 2) {
                float* tile = D + 32 * DP;                                   // [32][F] floats, row-major
                if (colok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        float2 v; v.x = o[0][r]; v.y = o[1][r];
                        *reinterpret_cast<float2*>(tile + rl * F + col) = v;
                    }
                }
                __syncthreads();
                const float inv_f = 1.0f / (float)Ft;
#pragma unroll
                for (int trip = 0; trip < ((NB_ABL & 2) ? 0 : 2); ++trip) {
                    float gy[4][4];
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4) {
                        const int rl = wave + 4 * (trip * 4 + u4);
                        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (okc && row0 + rl < n) t = *reinterpret_cast<const float4*>(tile + rl * F + j4);
                        gy[u4][0] = t.x; gy[u4][1] = t.y; gy[u4][2] = t.z; gy[u4][3] = t.w;
                    }
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4) {
                        const int u = trip * 4 + u4;
                        const int64_t rr = row0 + wave + 4 * u;
                        if (rr >= n) continue;                               // wave-uniform
                        float xh[4], g[4];
                        float a = 0.f, b = 0.f;
                        // (MSK: the per-element validity is recomputed here -- four registers less across the block loop, which is
                        // what keeps the <12, 2, true> instantiation inside the 256 registers of two workgroups per CU)
                        bool okl[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) okl[e] = MSK ? j4 + e < Ft : okc;
                        if constexpr (MSK) gte_ln_bwd_pre4m(gy[u4], zr[u], mu2[u], rs2[u], gam4, bet4, okl, lnb.relu, xh, g, a, b);
                        else gte_ln_bwd_pre4(gy[u4], zr[u], mu2[u], rs2[u], gam4, bet4, okc, lnb.relu, xh, g, a, b);
                        const float c1 = gte_group_sum<64>(a) * inv_f, c2 = gte_group_sum<64>(b) * inv_f;
                        float d[4];
                        if constexpr (MSK) gte_ln_bwd_post4m(g, xh, gam4, rs2[u], c1, c2, okl, d, c_dg, c_db, c_dz);
                        else gte_ln_bwd_post4(g, xh, gam4, rs2[u], c1, c2, okc, d, c_dg, c_db, c_dz);
                        if (okc && (!(NB_ABL & 8) || d[0] == 12345.f)) {
                            struct __attribute__((packed, aligned(4))) f4n { float x, y, z, w; };
                            f4n ov; ov.x = d[0]; ov.y = d[1]; ov.z = d[2]; ov.w = d[3];
                            *reinterpret_cast<f4n*>(dh + rr * lddh + j4) = ov;
                        }
                        if (imc && lnb.dzp3 && (!(NB_ABL & 8) || d[0] == 12345.f)) p3::store4(lnb.dzp3 + rr * lnb.ldp3, j4, d[0], d[1], d[2], d[3]);
                    }
                }
            } else
            if (colok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (rr < n) {
                        f2n v; v.x = o[0][r]; v.y = o[1][r];
                        *reinterpret_cast<f2n*>(dh + (int64_t)rr * lddh + col) = v;
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < ((NB_ABL & 1) ? 1 : 16); ++s) {
            const float a = i < KD ? D[(2 * s + hh) * DP + i] : 0.f;
            gw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, hv[s].x, gw[0], 0, 0, 0);
            gw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, hv[s].y, gw[1], 0, 0, 0);
        }
    }

    // the workgroup's partial: every wave writes its own column slice
    float* pp = partial + (int64_t)blockIdx.x * (2 * C * Ft + C);       // (compact: [2 C][Ft] + [C])
    if (colok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const int c = m < NCT ? m : m - NCT;
            if (m < KD && c < C) {
                float* dst = pp + (int64_t)((m < NCT ? 0 : C) + c) * Ft + col;
                if (!MSK || col < Ft) dst[0] = gw[0][r];
                if (!MSK || col + 1 < Ft) dst[1] = gw[1][r];
            }
        }
    }
    if (wave == 0 && i < C && hh == 0) pp[2 * C * Ft + i] = gb;
    if constexpr (LNB == 2) {                   // column partials: the four waves through LDS, added in wave order
        __syncthreads();
        float* red = D + 32 * DP;                // [4][3][F]
        if (okc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[(wave * 3 + 0) * F + j4 + e] = c_dg[e];
                red[(wave * 3 + 1) * F + j4 + e] = c_db[e];
                red[(wave * 3 + 2) * F + j4 + e] = c_dz[e];
            }
        }
        __syncthreads();
        float* lp = lnb.lnpart + (int64_t)blockIdx.x * 3 * Ft;           // (compact: [3][Ft])
        for (int t = threadIdx.x; t < 3 * F; t += 256) {
            const int qq = t / F, j = t - qq * F;
            if (!MSK || j < Ft) lp[qq * Ft + j] = ((red[t] + red[3 * F + t]) + red[6 * F + t]) + red[9 * F + t];
        }
    }
}

// dW[c, seg*F + k] = sum_b partial[b][(seg*C + c)*F + k];  dbias[c] = sum_b partial[b][2CF + c]   (fixed order)
// 16 slices of the block range per element (independent loads in flight), folded in slice order through LDS.
__global__ void __launch_bounds__(1024)
narrow_fold16_kernel(const float* __restrict__ partial, int nblk, int F, int C, float* __restrict__ dW, int64_t lddw,
                     float* __restrict__ dbias) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int total = 2 * C * F + C;
    const int e = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int b = sl;
        for (; b + 16 < nblk; b += 32) { s0 += partial[(int64_t)b * total + e]; s1 += partial[(int64_t)(b + 16) * total + e]; }
        if (b < nblk) s0 += partial[(int64_t)b * total + e];
    }
    part[sl][lane] = s0 + s1;
    __syncthreads();
    if (sl == 0 && e < total) {
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) v += part[t][lane];
        if (e < 2 * C * F) {
            const int c2 = e / F, k = e - c2 * F;
            const int seg = c2 / C, c = c2 - seg * C;
            dW[(int64_t)c * lddw + seg * F + k] = v;
        } else if (dbias) {
            dbias[e - 2 * C * F] = v;
        }
    }
}

// ---- fused head: logits = t_self + mean-aggregate(t_neigh), weighted cross-entropy, unnormalised gradient ------------
// Replaces three launches (9-wide aggregation, CE partial, CE gradient) by one.  4 lanes per node row (lane l owns columns
// l, l+4, l+8, l+12 < C <= 16), edges read four at a time and broadcast inside the group -- the same summation order as
// spmm_csr_kernel<F32, 4, 1, true>, so the logits are bitwise those of the unfused path.  Per row: arg-max (first maximum),
// log-sum-exp, nll = w_y (lse - z_y), and dl' = w_y (softmax - onehot) WITHOUT the 1 / sum(w): that factor needs all rows;
// it is linear, so the transpose aggregation of dl' and narrow_bwd_mfma_kernel (alpha) apply it later.
template <typename L>
__global__ void __launch_bounds__(256)
head_agg_ce_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ ew,
                   const float* __restrict__ t_neigh, int64_t ldn, float* __restrict__ logits, int64_t ldl,
                   const L* __restrict__ labels, const float* __restrict__ cw, int n, int C, int reduce,
                   float* __restrict__ dl, int64_t lddl, float* __restrict__ partial) {
    const int li = threadIdx.x & 3;
    const int r = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool row_ok = r < n;
    int lo = 0, hi = 0;
    if (row_ok) { lo = indptr[r]; hi = indptr[r + 1]; }
    const float scale = (reduce == GTE_REDUCE_MEAN) ? (hi > lo ? 1.0f / (float)(hi - lo) : 0.0f) : 1.0f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int eb = lo; eb < hi; eb += 4) {
        const int my_e = eb + li;
        int my_u = 0;
        float my_w = 0.f;
        if (my_e < hi) { my_u = indices[my_e]; my_w = ew ? ew[my_e] : 1.0f; }
        float v[4][4];
        float w[4];
        // lanes past the end of the row hold (source 0, weight 0): a valid address whose contribution is + 0 * x.
        // Quad broadcasts on DPP (a __shfl is a ds_bpermute: an LDS round trip per edge in the dependent chain)
        int u[4];
        u[0] = gte_quad_bcast<0>(my_u); u[1] = gte_quad_bcast<1>(my_u); u[2] = gte_quad_bcast<2>(my_u); u[3] = gte_quad_bcast<3>(my_u);
        w[0] = gte_quad_bcast<0>(my_w); w[1] = gte_quad_bcast<1>(my_w); w[2] = gte_quad_bcast<2>(my_w); w[3] = gte_quad_bcast<3>(my_w);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* xr = t_neigh + (int64_t)u[t] * ldn;
#pragma unroll
            for (int m = 0; m < 4; ++m) v[t][m] = (li + 4 * m < C) ? xr[li + 4 * m] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = fmaf(w[t], v[t][m], acc[m]);
    }
    float z[4];
    float zmax = -INFINITY;
    float* lrow = logits + (int64_t)(row_ok ? r : 0) * ldl;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int col = li + 4 * m;
        z[m] = -INFINITY;
        if (row_ok && col < C) {
            z[m] = lrow[col] + acc[m] * scale;
            lrow[col] = z[m];
            zmax = fmaxf(zmax, z[m]);
        }
    }
    zmax = fmaxf(zmax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, zmax), 0xB1, 0xf, 0xf, true)));
    zmax = fmaxf(zmax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, zmax), 0x4E, 0xf, 0xf, true)));
    int arg = 1 << 20;                                       // first column holding the maximum (torch.argmax)
    float se = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int col = li + 4 * m;
        if (row_ok && col < C) {
            if (z[m] == zmax) arg = min(arg, col);
            se += expf(z[m] - zmax);
        }
    }
    arg = min(arg, __builtin_amdgcn_update_dpp(0, arg, 0xB1, 0xf, 0xf, true));
    arg = min(arg, __builtin_amdgcn_update_dpp(0, arg, 0x4E, 0xf, 0xf, true));
    se = gte_group_sum<4>(se);
    const int y = row_ok ? gte_ce::label_of(labels, r) : -1;
    const bool yok = row_ok && y >= 0 && y < C;
    const float wy = yok ? (cw ? cw[y] : 1.0f) : 0.f;
    float zy = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) if (yok && li + 4 * m == y) zy = z[m];
    zy = gte_group_sum<4>(zy);
    if (row_ok && dl) {
        const float inv = 1.0f / se;
        float* drow = dl + (int64_t)r * lddl;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int col = li + 4 * m;
            if (col < C) drow[col] = wy * (expf(z[m] - zmax) * inv - (col == y ? 1.f : 0.f));
        }
    }
    // block partial {sum w nll, sum w, #correct}: one lane per row contributes; wave shuffle, then the 4 waves in order
    float loss = 0.f, wsum = 0.f, correct = 0.f;
    if (yok && li == 0) {
        loss = wy * (logf(se) + zmax - zy);
        wsum = wy;
        correct = (arg == y) ? 1.f : 0.f;
    }
    __shared__ float red[3][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        loss += __shfl_down(loss, off, 64);
        wsum += __shfl_down(wsum, off, 64);
        correct += __shfl_down(correct, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = loss; red[1][wave] = wsum; red[2][wave] = correct; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f, d = 0.f;
        for (int w = 0; w < 4; ++w) { a += red[0][w]; b += red[1][w]; d += red[2][w]; }
        partial[(int64_t)blockIdx.x * 3 + 0] = a;
        partial[(int64_t)blockIdx.x * 3 + 1] = b;
        partial[(int64_t)blockIdx.x * 3 + 2] = d;
    }
}

int narrow_blocks(int64_t n) {
    const int64_t b = gte::ceil_div(n, 4);
    return (int)(b < NB_MAX ? b : NB_MAX);
}
// workgroups of the matrix-pipe backward (each walks 32-row blocks): one per CU.  Every workgroup leaves an 18 KB dW
// partial behind (written here, read again by the fold): at 512 workgroups that was 14 MB of extra traffic on a 50 MB
// kernel; measured per step 0.773 (512) / 0.766 (384) / 0.761 (256) / 0.769 ms (192).
// per_cu = 2: the form with the LayerNorm backward behind the dh product (round 5: 256 registers, two workgroups per CU -- 38.6 -> 34.3 us at
// 24 k x 256 for +2.3 us of fold, profiles/r05/narrow_bwd_abl.txt)
int narrow_mfma_blocks(int64_t n, int per_cu = 1) {
    static const int forced = GTE_MEASURE_INT("GTE_NARROW_BLOCKS", 0);
    const int want = gte::device_props().cus * per_cu;
    const int cap = forced > 0 && forced <= NB_MAX ? forced : (want < NB_MAX ? want : NB_MAX);
    const int64_t b = gte::ceil_div(n, 32);
    return (int)(b < cap ? b : cap);
}

}  // namespace

// 16-row-block forward (narrow_fwd_mfma16_kernel): F a multiple of 16; GTE_NARROW_FWD16=0 keeps the 32-row kernel (A/B)
static bool narrow_fwd16(int64_t n_feat) {
    static const bool on = !GTE_MEASURE_OFF("GTE_NARROW_FWD16");
    return on && n_feat % 16 == 0;
}

extern "C" int gte_sage_narrow_supported(int64_t n_feat, int64_t n_out) {
    return (n_out >= 1 && n_out <= NC_MAX && n_feat >= 1 && n_feat <= 256) ? 1 : 0;
}

// n_true: the width of W's halves ([C][2 n_true]); n_feat >= n_true: the width computed over (h rows zero-padded up to it)
static int narrow_fwd_impl(const float* h, int64_t ldh, int64_t n_feat, int64_t n_true, const float* W, int64_t ldw,
                           const float* bias, int64_t n_out, float* t_self, int64_t ld_self, float* t_neigh,
                           int64_t ld_neigh, int64_t n_nodes, void* stream) {
    if (!gte_sage_narrow_supported(n_feat, n_out) || n_nodes < 0 || n_nodes > INT32_MAX || n_true < 1 || n_true > n_feat)
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_fwd: needs n_out <= 16 and n_feat <= 256");
    if (n_nodes == 0) return GTE_OK;
    if (!h || !W || !t_self || !t_neigh) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_fwd: null pointer");
    if (ldh < n_feat || ldw < 2 * n_true || ld_self < n_out || ld_neigh < n_out)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_fwd: leading dimension too small");
    if (n_true != n_feat && n_feat % 16 != 0)
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_fwd_pad: the padded width must be a multiple of 16");
    hipStream_t s = gte::as_stream(stream);
    if (n_feat % 8 == 0) {                                 // matrix-pipe version
        const int64_t nblk = gte::ceil_div(n_nodes, 32);
        // 4 waves per workgroup: every workgroup stages the W image (33 KB at F = 256) once, and that staging is most of a
        // workgroup's life at one row block per wave (measured at 24 k nodes: 4 waves 19.5 us, 2 waves 23.0, 1 wave 27.4)
        const int wpb = 4;
        const int mb = (int)(gte::ceil_div(nblk, wpb) < 2048 ? gte::ceil_div(nblk, wpb) : 2048);
        if (narrow_fwd16(n_feat)) {
            const int64_t nb16 = gte::ceil_div(gte::ceil_div(n_nodes, 16), 8);
            hipLaunchKernelGGL(narrow_fwd_mfma16_kernel<false>, dim3((unsigned)(nb16 < 2048 ? nb16 : 2048)), dim3(512),
                               (size_t)32 * (n_feat + 4) * 4, s, h, ldh, W, ldw, bias, t_self, ld_self, t_neigh, ld_neigh,
                               (int)n_nodes, (int)n_feat, (int)n_out, LnForward{}, (int)n_true);
            return gte::check_launch("sage_narrow_fwd");
        }
        hipLaunchKernelGGL(narrow_fwd_mfma_kernel<false>, dim3((unsigned)mb), dim3(64 * wpb), (size_t)32 * (n_feat + 4) * 4, s, h,
                           ldh, W, ldw, bias, t_self, ld_self, t_neigh, ld_neigh, (int)n_nodes, (int)n_feat, (int)n_out,
                           LnForward{}, (int)n_true);
        return gte::check_launch("sage_narrow_fwd");
    }
    const int blocks = (int)(gte::ceil_div(n_nodes, 4) < 2048 ? gte::ceil_div(n_nodes, 4) : 2048);
#define GTE_NF2(NJ, NCT)                                                                                                 \
    hipLaunchKernelGGL((narrow_fwd_kernel<NJ, NCT>), dim3((unsigned)blocks), dim3(256), 0, s, h, ldh, W, ldw, bias, t_self, \
                       ld_self, t_neigh, ld_neigh, (int)n_nodes, (int)n_feat, (int)n_out)
#define GTE_NF(NJ)                                                                               \
    if (n_out <= 4) GTE_NF2(NJ, 4); else if (n_out <= 8) GTE_NF2(NJ, 8); else if (n_out <= 12) GTE_NF2(NJ, 12); \
    else GTE_NF2(NJ, 16)
    if (n_feat <= 64) { GTE_NF(1); } else if (n_feat <= 128) { GTE_NF(2); } else { GTE_NF(4); }
#undef GTE_NF
#undef GTE_NF2
    return gte::check_launch("sage_narrow_fwd");
}

extern "C" int gte_sage_narrow_fwd(const float* h, int64_t ldh, int64_t n_feat, const float* W, int64_t ldw,
                                   const float* bias, int64_t n_out, float* t_self, int64_t ld_self, float* t_neigh,
                                   int64_t ld_neigh, int64_t n_nodes, void* stream) {
    return narrow_fwd_impl(h, ldh, n_feat, n_feat, W, ldw, bias, n_out, t_self, ld_self, t_neigh, ld_neigh, n_nodes, stream);
}
// ... on PADDED rows: h [n][n_pad] holds n_feat true columns followed by zeros (n_pad a multiple of 16, <= 256: how the one-call
// plan keeps hidden widths that are not a multiple of 16); W is the reference's [C][2 n_feat].  The products run over n_pad
// columns against a weight image whose columns n_feat .. n_pad-1 are zero: the same sums, on the matrix-pipe kernels.
extern "C" int gte_sage_narrow_pad_supported(int64_t n_feat, int64_t n_pad, int64_t n_out) {
    return (n_out >= 1 && n_out <= NC_MAX && n_feat >= 1 && n_feat <= n_pad && n_pad <= 256 && n_pad % 16 == 0) ? 1 : 0;
}
extern "C" int gte_sage_narrow_fwd_pad(const float* h, int64_t ldh, int64_t n_feat, int64_t n_pad, const float* W, int64_t ldw,
                                       const float* bias, int64_t n_out, float* t_self, int64_t ld_self, float* t_neigh,
                                       int64_t ld_neigh, int64_t n_nodes, void* stream) {
    if (!gte_sage_narrow_pad_supported(n_feat, n_pad, n_out))
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_fwd_pad: needs n_out <= 16, n_feat <= n_pad <= 256, n_pad %% 16 == 0");
    return narrow_fwd_impl(h, ldh, n_pad, n_feat, W, ldw, bias, n_out, t_self, ld_self, t_neigh, ld_neigh, n_nodes, stream);
}

extern "C" int gte_sage_narrow_fwd_ln_supported(int64_t n_feat, int64_t n_out) {
    return (gte_sage_narrow_supported(n_feat, n_out) && n_feat % 8 == 0) ? 1 : 0;
}

extern "C" int gte_sage_narrow_fwd_ln(const float* z, int64_t ldz, int64_t n_feat, const float* gamma, const float* beta,
                                      float eps, int relu, float* y, int64_t ldy, float* stats, const float* W, int64_t ldw,
                                      const float* bias, int64_t n_out, float* t_self, int64_t ld_self, float* t_neigh,
                                      int64_t ld_neigh, int64_t n_nodes, void* stream) {
    if (!gte_sage_narrow_fwd_ln_supported(n_feat, n_out) || n_nodes < 0 || n_nodes > INT32_MAX)
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_fwd_ln: needs n_out <= 16, n_feat <= 256, n_feat % 8 == 0");
    if (n_nodes == 0) return GTE_OK;
    if (!z || !gamma || !beta || !y || !W || !t_self || !t_neigh)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_fwd_ln: null pointer");
    if (ldz < n_feat || ldy < n_feat || ldw < 2 * n_feat || ld_self < n_out || ld_neigh < n_out)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_fwd_ln: leading dimension too small");
    const int64_t nblk = gte::ceil_div(n_nodes, 32);
    const int wpb = 4;
    const int mb = (int)(gte::ceil_div(nblk, wpb) < 2048 ? gte::ceil_div(nblk, wpb) : 2048);
    const LnForward lnf = {gamma, beta, eps, relu, y, ldy, stats};
    if (narrow_fwd16(n_feat)) {
        const int64_t nb16 = gte::ceil_div(gte::ceil_div(n_nodes, 16), 8);
        hipLaunchKernelGGL(narrow_fwd_mfma16_kernel<true>, dim3((unsigned)(nb16 < 2048 ? nb16 : 2048)), dim3(512),
                           (size_t)(32 * (n_feat + 4) + 2 * n_feat) * 4, gte::as_stream(stream), z, ldz, W, ldw, bias, t_self,
                           ld_self, t_neigh, ld_neigh, (int)n_nodes, (int)n_feat, (int)n_out, lnf, (int)n_feat);
        return gte::check_launch("sage_narrow_fwd_ln");
    }
    hipLaunchKernelGGL(narrow_fwd_mfma_kernel<true>, dim3((unsigned)mb), dim3(64 * wpb),
                       (size_t)(32 * (n_feat + 4) + 2 * n_feat) * 4, gte::as_stream(stream), z, ldz, W, ldw, bias, t_self, ld_self,
                       t_neigh, ld_neigh, (int)n_nodes, (int)n_feat, (int)n_out, lnf, (int)n_feat);
    return gte::check_launch("sage_narrow_fwd_ln");
}

extern "C" int64_t gte_sage_narrow_bwd_workspace_bytes(int64_t n_nodes, int64_t n_feat, int64_t n_out) {
    return gte::round_up((int64_t)narrow_blocks(n_nodes > 0 ? n_nodes : 1) * (2 * n_out * n_feat + n_out) * 4, 256);
}

namespace {
// n_true (0: = n_feat): the true width of W's halves / of the hidden rows; n_feat then is the padded width computed over
int narrow_bwd_impl(const float* dl, int64_t lddl, const float* q, int64_t ldq, const float* h, int64_t ldh, int64_t n_feat,
                    const float* W, int64_t ldw, int64_t n_out, float* dh, int64_t lddh, float* dW, int64_t lddw, float* dbias,
                    int64_t n_nodes, void* workspace, int64_t workspace_bytes, const float* ce_partial, int64_t ce_blocks,
                    float grad_scale, float* out3, void* stream, const LnBackward* lnb = nullptr, float* dgamma = nullptr,
                    float* dbeta = nullptr, float* dbias_below = nullptr, int64_t n_true = 0) {
    if (n_true <= 0) n_true = n_feat;
    if (!gte_sage_narrow_supported(n_feat, n_out) || n_nodes < 0 || n_nodes > INT32_MAX || n_true > n_feat)
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_bwd: needs n_out <= 16 and n_feat <= 256");
    if (n_nodes == 0) return GTE_OK;
    if (!dl || !q || !h || !W || !dW || !workspace)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_bwd: null pointer");
    if (lddl < n_out || ldq < n_out || ldh < n_feat || ldw < 2 * n_true || lddw < 2 * n_true || (dh && lddh < n_feat))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_bwd: leading dimension too small");
    if (workspace_bytes < gte_sage_narrow_bwd_workspace_bytes(n_nodes, n_feat, n_out))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "sage_narrow_bwd: workspace too small");
    const bool msk = n_true != n_feat;
    if (msk && (n_feat % 16 != 0 || !lnb))
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_bwd (padded rows): the padded width must be a multiple of 16, with the fused LayerNorm backward");
    hipStream_t s = gte::as_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    const int C = (int)n_out, F = (int)n_feat, Ft = (int)n_true;
    if (n_feat % 8 == 0) {                                 // matrix-pipe version
        const int nbm = narrow_mfma_blocks(n_nodes, lnb ? 2 : 1);
#define GTE_NBM(NCT, NCTM)                                                                                            \
    do {                                                                                                              \
        if (lnb && msk)                                                                                               \
            hipLaunchKernelGGL((narrow_bwd_mfma_kernel<NCTM, 2, true>), dim3((unsigned)nbm), dim3(256),               \
                               (size_t)(2 * NCTM * (F + 4) + 32 * (2 * NCTM + 1) + 32 * F) * 4, s, dl, lddl, q, ldq, h, ldh, W, ldw, dh, \
                               lddh, part, (int)n_nodes, F, C, ce_partial, ce_blocks, grad_scale, out3, *lnb, Ft);    \
        else if (lnb)                                                                                                 \
            hipLaunchKernelGGL((narrow_bwd_mfma_kernel<NCT, 2>), dim3((unsigned)nbm), dim3(256),                      \
                               (size_t)(2 * NCT * (F + 4) + 32 * (2 * NCT + 1) + 32 * F) * 4, s, dl, lddl, q, ldq, h, ldh, W, ldw, dh, \
                               lddh, part, (int)n_nodes, F, C, ce_partial, ce_blocks, grad_scale, out3, *lnb, 0);     \
        else                                                                                                          \
            hipLaunchKernelGGL((narrow_bwd_mfma_kernel<NCT, 0>), dim3((unsigned)nbm), dim3(256),                      \
                               (size_t)(2 * NCT * (F + 4) + 32 * (2 * NCT + 1)) * 4, s, dl, lddl, q, ldq, h, ldh, W, ldw, dh, lddh, \
                               part, (int)n_nodes, F, C, ce_partial, ce_blocks, grad_scale, out3, LnBackward{}, 0);   \
    } while (0)
        // (padded rows with 9 ... 12 classes take the 16-class instantiation: <12, 2, true> needs 259 registers -- 12 bytes of scratch per
        // lane inside the 256 of two workgroups per CU --, <16, 2, true> fits; eight more MFMAs per 32-row block)
        if (n_out <= 4) GTE_NBM(4, 4); else if (n_out <= 8) GTE_NBM(8, 8); else if (n_out <= 12) GTE_NBM(12, 16); else GTE_NBM(16, 16);
#undef GTE_NBM
        // (the partials are compact: rows of Ft = the true width)
        if (lnb) {                                          // column sums of the fused LayerNorm backward
            const bool own = !gte::defer_fold(lnb->lnpart, 3 * (int64_t)Ft, nbm, 1, Ft, dgamma, Ft);
            if (own) {                                      // no deferral open: run the three folds now, as one launch
                int rc = gte_fold_defer_begin(stream);
                if (rc != GTE_OK) return rc;
                gte::defer_fold(lnb->lnpart, 3 * (int64_t)Ft, nbm, 1, Ft, dgamma, Ft);
            }
            gte::defer_fold(lnb->lnpart + Ft, 3 * (int64_t)Ft, nbm, 1, Ft, dbeta, Ft);
            gte::defer_fold(lnb->lnpart + 2 * Ft, 3 * (int64_t)Ft, nbm, 1, Ft, dbias_below, Ft);
            if (own) {
                // the narrow layer's own folds join the same launch
                const int64_t ps = 2 * (int64_t)C * Ft + C;
                gte::defer_fold(part, ps, nbm, C, Ft, dW, lddw);
                gte::defer_fold(part + (int64_t)C * Ft, ps, nbm, C, Ft, dW + Ft, lddw);
                gte::defer_fold(part + 2 * (int64_t)C * Ft, ps, nbm, 1, C, dbias, C);
                return gte_fold_defer_flush();
            }
        }
        const int64_t pstride = 2 * (int64_t)C * Ft + C;
        if (gte::defer_fold(part, pstride, nbm, C, Ft, dW, lddw)) {                    // [dl^T h]
            gte::defer_fold(part + (int64_t)C * Ft, pstride, nbm, C, Ft, dW + Ft, lddw);   // [q^T h]
            gte::defer_fold(part + 2 * (int64_t)C * Ft, pstride, nbm, 1, C, dbias, C);
        } else {
            hipLaunchKernelGGL(narrow_fold16_kernel, dim3((unsigned)gte::ceil_div(2 * C * Ft + C, 64)), dim3(1024), 0, s, part, nbm,
                               Ft, C, dW, lddw, dbias);
        }
        return gte::check_launch("sage_narrow_bwd");
    }
    if (ce_partial || lnb) return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_bwd_ce / _ln: needs n_feat %% 8 == 0");
    const int nb = narrow_blocks(n_nodes);
#define GTE_NB2(NJ, NCT)                                                                                                  \
    hipLaunchKernelGGL((narrow_bwd_kernel<NJ, NCT>), dim3((unsigned)nb), dim3(256), (size_t)(2 * C * 64 * NJ + C) * 4, s, \
                       dl, lddl, q, ldq, h, ldh, W, ldw, dh, lddh, part, (int)n_nodes, F, C)
#define GTE_NB(NJ)                                                                               \
    if (n_out <= 4) GTE_NB2(NJ, 4); else if (n_out <= 8) GTE_NB2(NJ, 8); else if (n_out <= 12) GTE_NB2(NJ, 12); \
    else GTE_NB2(NJ, 16)
    if (n_feat <= 64) { GTE_NB(1); } else if (n_feat <= 128) { GTE_NB(2); } else { GTE_NB(4); }
#undef GTE_NB
#undef GTE_NB2
    hipLaunchKernelGGL(narrow_fold_kernel, dim3((unsigned)gte::ceil_div(2 * C * F + C, 64)), dim3(256), 0, s, part, nb, F, C,
                       dW, lddw, dbias);
    return gte::check_launch("sage_narrow_bwd");
}
}  // namespace

extern "C" int gte_sage_narrow_bwd(const float* dl, int64_t lddl, const float* q, int64_t ldq, const float* h,
                                   int64_t ldh, int64_t n_feat, const float* W, int64_t ldw, int64_t n_out, float* dh,
                                   int64_t lddh, float* dW, int64_t lddw, float* dbias, int64_t n_nodes,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    return narrow_bwd_impl(dl, lddl, q, ldq, h, ldh, n_feat, W, ldw, n_out, dh, lddh, dW, lddw, dbias, n_nodes, workspace,
                           workspace_bytes, nullptr, 0, 1.f, nullptr, stream);
}

extern "C" int gte_head_supported(int64_t n_feat, int64_t n_classes) {
    return (gte_sage_narrow_supported(n_feat, n_classes) && n_feat % 8 == 0) ? 1 : 0;
}

extern "C" int64_t gte_head_agg_ce_workspace_bytes(int64_t n_nodes) {
    return gte::round_up(gte::ceil_div(n_nodes > 0 ? n_nodes : 1, 64) * 3 * (int64_t)sizeof(float), 256);
}

extern "C" int gte_head_agg_ce(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* t_neigh,
                               int64_t ld_neigh, float* logits, int64_t ld_logits, const void* labels, int labels_f32,
                               const float* class_weight, int64_t n_nodes, int64_t n_classes, int reduce, float* dl_unscaled,
                               int64_t lddl, void* ce_partial, int64_t ce_partial_bytes, void* stream) {
    if (n_nodes <= 0 || n_nodes > INT32_MAX || n_classes < 1 || n_classes > NC_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "head_agg_ce: needs 1 <= n_classes <= 16 and n_nodes >= 1");
    if (!indptr || !t_neigh || !logits || !labels || !ce_partial)          // indices may be NULL for an edgeless graph
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "head_agg_ce: null pointer");
    if (ld_neigh < n_classes || ld_logits < n_classes || (dl_unscaled && lddl < n_classes))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "head_agg_ce: leading dimension too small");
    if (reduce != GTE_REDUCE_SUM && reduce != GTE_REDUCE_MEAN)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "head_agg_ce: reduce must be 0 (sum) or 1 (mean)");
    if (ce_partial_bytes < gte_head_agg_ce_workspace_bytes(n_nodes))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "head_agg_ce: partial buffer too small");
    hipStream_t s = gte::as_stream(stream);
    const unsigned nb = (unsigned)gte::ceil_div(n_nodes, 64);
    float* part = reinterpret_cast<float*>(ce_partial);
    if (labels_f32)
        hipLaunchKernelGGL(head_agg_ce_kernel<float>, dim3(nb), dim3(256), 0, s, indptr, indices, eweight, t_neigh, ld_neigh,
                           logits, ld_logits, (const float*)labels, class_weight, (int)n_nodes, (int)n_classes, reduce,
                           dl_unscaled, lddl, part);
    else
        hipLaunchKernelGGL(head_agg_ce_kernel<int64_t>, dim3(nb), dim3(256), 0, s, indptr, indices, eweight, t_neigh, ld_neigh,
                           logits, ld_logits, (const int64_t*)labels, class_weight, (int)n_nodes, (int)n_classes, reduce,
                           dl_unscaled, lddl, part);
    return gte::check_launch("head_agg_ce");
}

extern "C" int gte_sage_narrow_bwd_ce(const float* dl_unscaled, int64_t lddl, const float* q_unscaled, int64_t ldq,
                                      const float* h, int64_t ldh, int64_t n_feat, const float* W, int64_t ldw, int64_t n_out,
                                      float* dh, int64_t lddh, float* dW, int64_t lddw, float* dbias, int64_t n_nodes,
                                      void* workspace, int64_t workspace_bytes, const void* ce_partial, float grad_scale,
                                      float* out3, void* stream) {
    if (!ce_partial || !out3) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_bwd_ce: null pointer");
    if (!gte_head_supported(n_feat, n_out)) return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_bwd_ce: see gte_head_supported");
    return narrow_bwd_impl(dl_unscaled, lddl, q_unscaled, ldq, h, ldh, n_feat, W, ldw, n_out, dh, lddh, dW, lddw, dbias, n_nodes,
                           workspace, workspace_bytes, reinterpret_cast<const float*>(ce_partial),
                           gte::ceil_div(n_nodes > 0 ? n_nodes : 1, 64), grad_scale, out3, stream);
}


// gte_sage_narrow_bwd_ln in the row form: the dh tile of a row block goes through LDS and whole rows get the LayerNorm(+ReLU)
// backward with 16-byte accesses; dz_below as fp32 AND as a P3 image (dzp3 nullable).  Bit for bit
// gte_sage_narrow_bwd[_ce] + gte_ln_relu_bwd_p3; one launch and the [N, F] round trip of dh less.  n_feat % 4 == 0.
extern "C" int gte_sage_narrow_bwd_ln_p3(const float* dl, int64_t lddl, const float* q, int64_t ldq, const float* h, int64_t ldh,
                                         int64_t n_feat, const float* W, int64_t ldw, int64_t n_out, float* dz_below, int64_t lddz,
                                         void* dzp3, int64_t ldp3, float* dW, int64_t lddw, float* dbias, int64_t n_nodes,
                                         void* workspace, int64_t workspace_bytes, const void* ce_partial, float grad_scale,
                                         float* out3, const float* z_below, int64_t ldz, const float* stats_below,
                                         const float* gamma_below, const float* beta_below, int relu_below, float* dgamma_below,
                                         float* dbeta_below, float* dbias_below, void* ln_workspace, int64_t ln_workspace_bytes,
                                         void* stream) {
    if (!gte_head_supported(n_feat, n_out)) return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_bwd_ln_p3: see gte_head_supported");
    if (!dz_below || !z_below || !stats_below || !gamma_below || !beta_below || !ln_workspace || (ce_partial && !out3))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_bwd_ln_p3: null pointer");
    if (ldz < n_feat || lddz < n_feat || (dzp3 && (ldp3 < (int64_t)p3::row_bytes(n_feat) || ldp3 % 16 != 0 || n_feat % 16 != 0)))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_bwd_ln_p3: leading dimension too small (the image needs n_feat %% 16 == 0)");
    if (ln_workspace_bytes < gte_sage_narrow_bwd_ln_workspace_bytes(n_nodes, n_feat))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "sage_narrow_bwd_ln_p3: LayerNorm workspace too small");
    const LnBackward lnb = {z_below, ldz, stats_below, gamma_below, beta_below, relu_below, reinterpret_cast<float*>(ln_workspace),
                            reinterpret_cast<char*>(dzp3), ldp3};
    return narrow_bwd_impl(dl, lddl, q, ldq, h, ldh, n_feat, W, ldw, n_out, dz_below, lddz, dW, lddw, dbias, n_nodes, workspace,
                           workspace_bytes, reinterpret_cast<const float*>(ce_partial),
                           gte::ceil_div(n_nodes > 0 ? n_nodes : 1, 64), grad_scale, out3, stream, &lnb, dgamma_below,
                           dbeta_below, dbias_below);
}

// gte_sage_narrow_bwd_ln_p3 on PADDED rows (gte_sage_narrow_fwd_pad): h / z_below / dz_below rows hold n_feat true columns and
// zeros up to n_pad (a multiple of 16; the image dzp3 is ceil(n_feat / 16) = n_pad / 16 blocks wide); W, dW are the reference's
// [C][2 n_feat]; the LayerNorm backward runs over the n_feat true columns (the arithmetic of gte_ln_relu_bwd_p3 at that width) and
// writes zeros into the padding.  Workspaces as for n_pad.
extern "C" int gte_sage_narrow_bwd_ln_p3_pad(const float* dl, int64_t lddl, const float* q, int64_t ldq, const float* h, int64_t ldh,
                                             int64_t n_feat, int64_t n_pad, const float* W, int64_t ldw, int64_t n_out, float* dz_below,
                                             int64_t lddz, void* dzp3, int64_t ldp3, float* dW, int64_t lddw, float* dbias,
                                             int64_t n_nodes, void* workspace, int64_t workspace_bytes, const void* ce_partial,
                                             float grad_scale, float* out3, const float* z_below, int64_t ldz, const float* stats_below,
                                             const float* gamma_below, const float* beta_below, int relu_below, float* dgamma_below,
                                             float* dbeta_below, float* dbias_below, void* ln_workspace, int64_t ln_workspace_bytes,
                                             void* stream) {
    if (!gte_sage_narrow_pad_supported(n_feat, n_pad, n_out))
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_narrow_bwd_ln_p3_pad: needs n_out <= 16, n_feat <= n_pad <= 256, n_pad %% 16 == 0");
    if (!dz_below || !z_below || !stats_below || !gamma_below || !beta_below || !ln_workspace || (ce_partial && !out3))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_bwd_ln_p3_pad: null pointer");
    if (ldz < n_pad || lddz < n_pad || (dzp3 && (ldp3 < (int64_t)p3::row_bytes(n_pad) || ldp3 % 16 != 0)))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_narrow_bwd_ln_p3_pad: leading dimension too small");
    if (ln_workspace_bytes < gte_sage_narrow_bwd_ln_workspace_bytes(n_nodes, n_pad))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "sage_narrow_bwd_ln_p3_pad: LayerNorm workspace too small");
    const LnBackward lnb = {z_below, ldz, stats_below, gamma_below, beta_below, relu_below, reinterpret_cast<float*>(ln_workspace),
                            reinterpret_cast<char*>(dzp3), ldp3};
    return narrow_bwd_impl(dl, lddl, q, ldq, h, ldh, n_pad, W, ldw, n_out, dz_below, lddz, dW, lddw, dbias, n_nodes, workspace,
                           workspace_bytes, reinterpret_cast<const float*>(ce_partial),
                           gte::ceil_div(n_nodes > 0 ? n_nodes : 1, 64), grad_scale, out3, stream, &lnb, dgamma_below,
                           dbeta_below, dbias_below, n_feat);
}

extern "C" int64_t gte_sage_narrow_bwd_ln_workspace_bytes(int64_t n_nodes, int64_t n_feat) {
    return gte::round_up((int64_t)narrow_mfma_blocks(n_nodes > 0 ? n_nodes : 1, 2) * 3 * n_feat * 4, 256);
}
