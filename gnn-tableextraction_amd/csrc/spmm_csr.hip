// Neighbour aggregation for page graphs: CSR gather-SpMM on gfx950 (wave64).
//
//   out[v, :] = scale_v * sum_{e in row v} w[e] * x[indices[e], :]
//
// replaces the reference's DGL call  g.update_all(fn.u_mul_e('h','feat','m'), fn.sum('m','h'))
// (src/components/graphs/models.py:53-54), the fn.mean reducer (:146-149) and get_norm (:74-78).
// Run on the out-edge CSR it is that call's backward (DGL GSpMM.backward).
//
// Roofline: HBM.  Algorithmic bytes per destination row = 2*F*s + 8*deg + 4.
//
// Mapping (CDNA4):
//   * a group of G lanes (G = 4..64, power of two, chosen from F) owns one destination row;
//     a wave64 therefore processes 64/G rows at once -- small F (13, 9) does not idle lanes.
//   * every lane owns up to CPL 16-byte chunks of the feature row (global_load_dwordx4 through
//     4-byte-aligned vector types: gfx950 takes unaligned wide loads, so F = 831 rows use the
//     same instructions as 16-byte-aligned ones) plus, for the
//     F mod 4 tail, one scalar element on the first lanes of the group.
//   * the row's edge list (indices, weights) is read ONCE, coalesced, one edge per lane, and
//     broadcast inside the group with ds_bpermute (__shfl); four source rows are in flight
//     per lane-group per iteration to cover HBM/L2 latency.
//   * fixed summation order (CSR order) => bit-reproducible run to run; fp32 accumulate.
//   * blocks take contiguous row ranges and blockIdx is remapped so each XCD's L2 serves one
//     contiguous 1/8 of the (reading-order) node range: neighbouring rows share sources.
#include "gte_common.h"
#include "p3.h"

#include <stdlib.h>
#include <type_traits>

namespace {

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(2))) u4u { unsigned x, y, z, w; };

// ---- element traits: f32 (4 per 16-byte chunk) and bf16 (8 per chunk, f32 accumulate) ----
struct F32 {
    using elem = float;
    static constexpr int EPC = 4;
    static __device__ __forceinline__ void load(const elem* p, float (&v)[4]) {
        const f4u t = *reinterpret_cast<const f4u*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(elem* p, const float (&v)[4]) {
        f4u t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        *reinterpret_cast<f4u*>(p) = t;
    }
    static __device__ __forceinline__ float to_f32(elem e) { return e; }
    static __device__ __forceinline__ elem from_f32(float f) { return f; }
};

struct BF16 {
    using elem = unsigned short;
    static constexpr int EPC = 8;
    static __device__ __forceinline__ float to_f32(elem e) { return __uint_as_float(((unsigned)e) << 16); }
    static __device__ __forceinline__ elem from_f32(float f) {
        // round-to-nearest-even via the hardware convert (keeps NaN a NaN on gfx950)
        __bf16 b = (__bf16)f;
        return *reinterpret_cast<unsigned short*>(&b);
    }
    static __device__ __forceinline__ void load(const elem* p, float (&v)[8]) {
        unsigned w[4];
        const u4u t = *reinterpret_cast<const u4u*>(p);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(elem* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (unsigned)from_f32(v[2 * i]) | ((unsigned)from_f32(v[2 * i + 1]) << 16);
        u4u t; t.x = w[0]; t.y = w[1]; t.z = w[2]; t.w = w[3];
        *reinterpret_cast<u4u*>(p) = t;
    }
};

constexpr int kEdgeUnroll = 4;   // source rows in flight per lane group

// Optional LayerNorm(+ReLU) epilogue (LNE kernels: f32, accumulate, the whole row inside one lane group): the row a
// group has just finished (z = out + aggregate) is normalised in registers and written to y as well -- the separate
// LayerNorm kernel would read z back from HBM (models.py:64-66 after :53-54 in transform-then-aggregate order).
struct LnEpilogue {
    const float* gamma; const float* beta; float eps; int relu;
    float* y; int64_t ldy; float* stats;                 // stats[row] = mean, stats[n_rows + row] = rstd
    char* yp3; int64_t ldyp3;                            // optional P3 image of y (csrc/p3.h) for the next layer's planes GEMM
    int n_true;                                          // LayerNorm width: the kernel processes n_feat >= n_true columns (the rows are
                                                         // padded to a multiple of 4 / 16 floats), columns >= n_true are written as zeros
};

template <int G>
__device__ __forceinline__ float group_sum(float v) { return gte_group_sum<G>(v); }

// MASK, LNE kernels: the LayerNorm width ln.n_true is smaller than the n_feat columns processed (padded rows): per-element masks.
// Without it the epilogue is the round-3 code, instruction for instruction (24.5 us at 24 k x 256; the masks cost 6 us there).
// MASK, P3-output kernels: n_feat is not a multiple of 16 (tail elements collected by lane 0, zero quarter blocks).
#ifndef GTE_SPMM_NO_PREFETCH
#define GTE_SPMM_NO_PREFETCH 0     // measurement build: 1 = the self term loaded behind the edge loop (round 3's order)
#endif
template <typename T, int G, int CPL, bool ACCUM, bool LNE = false, bool MASK = false>
__global__ void __launch_bounds__(256)
spmm_csr_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                const float* __restrict__ ew, const typename T::elem* __restrict__ x, int64_t ldx,
                typename T::elem* __restrict__ out, int64_t ldo, int n_rows, int n_feat, int reduce,
                int rows_per_block, const LnEpilogue ln = LnEpilogue{}, char* __restrict__ outp3 = nullptr, int64_t ldp3 = 0) {
    using elem = typename T::elem;
    constexpr int EPC = T::EPC;
    // source rows in flight per lane group: four; two where a lane owns four chunks AND carries the LayerNorm epilogue (rows up to
    // 1024 wide: 179 VGPRs -> two waves per SIMD with four in flight)
    constexpr int EU = (LNE && CPL >= 4) ? 2 : kEdgeUnroll;
    constexpr int RPW = gte::kWave / G;                 // rows per wave per pass
    const int lane = threadIdx.x & (gte::kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int sub = lane / G, li = lane % G;
    const int nchunk = n_feat / EPC;                    // full 16-byte chunks per row
    const int rem = n_feat - nchunk * EPC;              // scalar tail elements (< EPC <= G... EPC 8 needs G >= 8)
    const unsigned lb = gte_xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (int)lb * rows_per_block;
    const int row_end = min(row0 + rows_per_block, n_rows);

    for (int rbase = row0 + wave * RPW; rbase < row_end; rbase += 4 * RPW) {
        const int r = rbase + sub;
        const bool row_ok = r < row_end;
        int lo = 0, hi = 0;
        if constexpr (RPW <= 4) {
            // few rows per wave: their indptr pairs through the scalar cache (uniform addresses -> s_load), the first link
            // of the row's dependent chain indptr -> edge list -> source rows
            const int rb_u = __builtin_amdgcn_readfirstlane(rbase);
#pragma unroll
            for (int s2 = 0; s2 < RPW; ++s2) {
                int l = 0, h = 0;
                if (rb_u + s2 < row_end) { l = indptr[rb_u + s2]; h = indptr[rb_u + s2 + 1]; }
                if (sub == s2) { lo = l; hi = h; }
            }
        } else {
            if (row_ok) { lo = indptr[r]; hi = indptr[r + 1]; }
        }
        const float scale = (reduce == GTE_REDUCE_MEAN) ? (hi > lo ? 1.0f / (float)(hi - lo) : 0.0f) : 1.0f;

        for (int cb = 0; cb < nchunk || (cb == 0 && rem); cb += G * CPL) {
            float acc[CPL][EPC];
#pragma unroll
            for (int j = 0; j < CPL; ++j)
#pragma unroll
                for (int q = 0; q < EPC; ++q) acc[j][q] = 0.f;
            float tail = 0.f;
            const bool do_tail = (cb == 0) && (li < rem);
            const int64_t tail_off = (int64_t)nchunk * EPC + li;
            // ACCUM: the row's own chunks of `out` (the self term of a transform-first layer) are requested HERE, in front of the
            // edge loop -- they depend on nothing, and behind the loop they were one more memory round trip on the row's chain
            float opre[ACCUM ? CPL : 1][EPC];
            if constexpr (ACCUM && !GTE_SPMM_NO_PREFETCH) {
                if (row_ok) {
                    const elem* orow_c = out + (int64_t)r * ldo;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int c = cb + li + j * G;
                        if (c < nchunk) T::load(orow_c + (int64_t)c * EPC, opre[j]);
                    }
                }
            }

            for (int eb = lo; eb < hi; eb += G) {
                const int my_e = eb + li;
                int my_u = 0;
                float my_w = 0.f;
                if (my_e < hi) {
                    my_u = indices[my_e];
                    my_w = ew ? ew[my_e] : 1.0f;
                }
                const int cnt = min(G, hi - eb);
                for (int t = 0; t < cnt; t += EU) {
                    int u[EU];
                    float w[EU];
                    if constexpr (G == 4 && EU == 4) {
                        // one quad per row: DPP quad broadcasts (lanes past the row end hold source 0 / weight 0)
                        u[0] = gte_quad_bcast<0>(my_u); u[1] = gte_quad_bcast<1>(my_u);
                        u[2] = gte_quad_bcast<2>(my_u); u[3] = gte_quad_bcast<3>(my_u);
                        w[0] = gte_quad_bcast<0>(my_w); w[1] = gte_quad_bcast<1>(my_w);
                        w[2] = gte_quad_bcast<2>(my_w); w[3] = gte_quad_bcast<3>(my_w);
                    } else {
#pragma unroll
                    for (int k = 0; k < EU; ++k) {
                        // past the end of the row: re-use the last valid source (cache hit) with w = 0
                        const int tt = min(t + k, cnt - 1);
                        u[k] = __shfl(my_u, tt, G);
                        w[k] = (t + k < cnt) ? __shfl(my_w, tt, G) : 0.f;
                    }
                    }
                    float v[EU][CPL][EPC];
                    float tv[EU];
#pragma unroll
                    for (int k = 0; k < EU; ++k) {
                        const elem* xr = x + (int64_t)u[k] * ldx;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) {
                            const int c = cb + li + j * G;
                            if (c < nchunk) T::load(xr + (int64_t)c * EPC, v[k][j]);
                        }
                        tv[k] = do_tail ? T::to_f32(xr[tail_off]) : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < EU; ++k) {
#pragma unroll
                        for (int j = 0; j < CPL; ++j) {
                            const int c = cb + li + j * G;
                            if (c < nchunk) {
#pragma unroll
                                for (int q = 0; q < EPC; ++q) acc[j][q] = fmaf(w[k], v[k][j][q], acc[j][q]);
                            }
                        }
                        tail = fmaf(w[k], tv[k], tail);
                    }
                }
            }

            // P3 output of a width that is not a multiple of 4: the (< 4) tail elements sit one per lane on the first lanes of the
            // group; lane 0 collects them into the row's last quarter block (lanes >= rem hold tail = 0)
            float tq[3] = {0.f, 0.f, 0.f};
            if constexpr (std::is_same<T, F32>::value && !ACCUM && !LNE && MASK) {
                if (outp3 && cb == 0 && rem > 0) {                 // (uniform: widths that are a multiple of 4 skip the shuffles)
                    const float ts = tail * scale;
                    tq[0] = __shfl(ts, 0, G); tq[1] = __shfl(ts, 1, G); tq[2] = __shfl(ts, 2, G);
                }
            }
            if (row_ok) {
                elem* orow = out + (int64_t)r * ldo;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int c = cb + li + j * G;
                    if (c < nchunk) {
                        float o[EPC];
                        if constexpr (ACCUM) {
                            if constexpr (GTE_SPMM_NO_PREFETCH) T::load(orow + (int64_t)c * EPC, o);
                            else {
#pragma unroll
                                for (int q = 0; q < EPC; ++q) o[q] = opre[j][q];
                            }
#pragma unroll
                            for (int q = 0; q < EPC; ++q) o[q] += acc[j][q] * scale;
                            if constexpr (LNE && MASK) {     // columns past the LayerNorm width are padding: zero
#pragma unroll
                                for (int q = 0; q < EPC; ++q) if (c * EPC + q >= ln.n_true) o[q] = 0.f;
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < EPC; ++q) o[q] = acc[j][q] * scale;
                        }
                        if constexpr (std::is_same<T, F32>::value && !ACCUM && !LNE) {
                            // the result as a P3 image (the operand of a planes GEMM) instead of fp32
                            if (outp3) p3::store4(outp3 + (int64_t)r * ldp3, c * 4, o[0], o[1], o[2], o[3]);
                            else T::store(orow + (int64_t)c * EPC, o);
                        } else {
                            T::store(orow + (int64_t)c * EPC, o);
                        }
                        if constexpr (LNE) {
#pragma unroll
                            for (int q = 0; q < EPC; ++q) acc[j][q] = o[q];          // keep z for the epilogue
                        }
                    } else if constexpr (LNE) {
#pragma unroll
                        for (int q = 0; q < EPC; ++q) acc[j][q] = 0.f;
                    }
                }
                if constexpr (LNE) {                         // rem == 0, nchunk <= G * CPL: acc[j] are this lane's z chunks (padding = 0)
                    const int nt = ln.n_true;
                    float sm = 0.f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j)
#pragma unroll
                        for (int q = 0; q < EPC; ++q) sm += acc[j][q];
                    const float mean = group_sum<G>(sm) / (float)nt;
                    float sq = 0.f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j)
#pragma unroll
                        for (int q = 0; q < EPC; ++q) {
                            const bool in = MASK ? ((li + j * G) * EPC + q < nt) : (li + j * G < nchunk);
                            const float d = in ? acc[j][q] - mean : 0.f;
                            sq = fmaf(d, d, sq);
                        }
                    const float rstd = rsqrtf(group_sum<G>(sq) / (float)nt + ln.eps);
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int c = li + j * G;
                        if (c < nchunk) {
                            float yv[EPC];
#pragma unroll
                            for (int q = 0; q < EPC; ++q) {
                                const int col = c * EPC + q;
                                if constexpr (MASK) {
                                    const bool ok = col < nt;
                                    float v = fmaf((acc[j][q] - mean) * rstd, ok ? ln.gamma[col] : 0.f, ok ? ln.beta[col] : 0.f);
                                    v = ln.relu ? fmaxf(v, 0.f) : v;
                                    yv[q] = ok ? v : 0.f;
                                } else {
                                    const float v = fmaf((acc[j][q] - mean) * rstd, ln.gamma[col], ln.beta[col]);
                                    yv[q] = ln.relu ? fmaxf(v, 0.f) : v;
                                }
                            }
                            if (ln.y) F32::store(ln.y + (int64_t)r * ln.ldy + (int64_t)c * EPC, yv);
                            if constexpr (EPC == 4) {
                                if (ln.yp3) p3::store4(ln.yp3 + (int64_t)r * ln.ldyp3, c * 4, yv[0], yv[1], yv[2], yv[3]);
                            }
                        }
                    }
                    if (li == 0 && ln.stats) { ln.stats[r] = mean; ln.stats[n_rows + r] = rstd; }
                }
                if constexpr (std::is_same<T, F32>::value && !ACCUM && !LNE && MASK) {
                    // P3 output: the quarter blocks past the last full chunk up to the image's 16-column block boundary -- the
                    // tail elements (lane 0) and zeros
                    if (outp3 && cb == 0 && li < 4 && (n_feat & 15) != 0) {
                        const int qi = nchunk + li;
                        if (qi * 4 < (int)p3::blocks(n_feat) * p3::BLOCK) {
                            const bool t = li == 0 && rem > 0;
                            p3::store4(outp3 + (int64_t)r * ldp3, qi * 4, t ? tq[0] : 0.f, t ? tq[1] : 0.f, t ? tq[2] : 0.f, 0.f);
                        }
                    }
                }
                if (do_tail && !(std::is_same<T, F32>::value && !ACCUM && !LNE && MASK && outp3)) {
                    float o = tail * scale;
                    if constexpr (ACCUM) o += T::to_f32(orow[tail_off]);
                    orow[tail_off] = T::from_f32(o);
                }
            }
        }
    }
}

template <typename T, int G, int CPL, bool ACCUM, bool MASK = false>
int launch_g(const int32_t* indptr, const int32_t* indices, const float* ew, const void* x, int64_t ldx,
             void* out, int64_t ldo, int64_t n_rows, int64_t n_feat, int reduce, hipStream_t s, char* outp3 = nullptr,
             int64_t ldp3 = 0) {
    using elem = typename T::elem;
    constexpr int RPW = gte::kWave / G;
    // 4 waves x RPW rows x `passes` passes, contiguous.  A wave walks its passes one after the other, each a chain of
    // dependent index -> row latencies, so passes only pay when the grid already oversubscribes the chip: 16 blocks
    // per CU before a second pass (measured at 21.5 k rows: F = 256 16.8 -> 14.3 us, F = 831 45 -> 39 us; narrow rows
    // at 4 passes had left two thirds of the CUs empty: 13 -> 6 us).
    int passes = 4;
    while (passes > 1 && gte::ceil_div(n_rows, (int64_t)4 * RPW * passes) < (int64_t)16 * gte::device_props().cus) passes /= 2;
    const int rows_per_block = 4 * RPW * passes;
    const int64_t nblocks = gte::ceil_div(n_rows, rows_per_block);
    dim3 grid((unsigned)nblocks), block(256);
    hipLaunchKernelGGL((spmm_csr_kernel<T, G, CPL, ACCUM, false, MASK>), grid, block, 0, s, indptr, indices, ew,
                       (const elem*)x, ldx, (elem*)out, ldo, (int)n_rows, (int)n_feat, reduce, rows_per_block, LnEpilogue{}, outp3,
                       ldp3);
    return gte::check_launch("spmm_csr");
}

template <typename T, bool ACCUM>
int dispatch(const int32_t* indptr, const int32_t* indices, const float* ew, const void* x, int64_t ldx,
             void* out, int64_t ldo, int64_t n_rows, int64_t n_feat, int reduce, hipStream_t s, char* outp3 = nullptr,
             int64_t ldp3 = 0) {
    const int64_t nchunk = n_feat / T::EPC;
    // (an image output of a width that is not a multiple of 16 runs the instantiation with the tail / zero-block code)
    const bool tail_img = std::is_same<T, F32>::value && !ACCUM && outp3 && (n_feat % 16) != 0;
#define GTE_L(G, CPL)                                                                                                        \
    do {                                                                                                                     \
        if constexpr (std::is_same<T, F32>::value && !ACCUM) {                                                               \
            if (tail_img) return launch_g<T, G, CPL, ACCUM, true>(indptr, indices, ew, x, ldx, out, ldo, n_rows, n_feat, reduce, s, outp3, ldp3); \
        }                                                                                                                    \
        return launch_g<T, G, CPL, ACCUM>(indptr, indices, ew, x, ldx, out, ldo, n_rows, n_feat, reduce, s, outp3, ldp3);   \
    } while (0)
    if (nchunk <= 8 && T::EPC <= 8) {
        if (nchunk <= 4 && T::EPC == 4) GTE_L(4, 1);
        GTE_L(8, 1);
    }
    if (nchunk <= 16) GTE_L(16, 1);
    if (nchunk <= 32) GTE_L(32, 1);
    if (nchunk <= 64) GTE_L(32, 2);                   // two rows per wave, two chunks per lane: 14.1 vs 15.2 us (G = 64) at 21.5 k x 256
    if (nchunk <= 128) GTE_L(64, 2);
    GTE_L(64, 4);                                     // wider rows loop over 256-chunk feature blocks
#undef GTE_L
}

template <int G, int CPL = 1, bool MASK = false>
int launch_ln(const int32_t* indptr, const int32_t* indices, const float* ew, const float* x, int64_t ldx, float* out,
              int64_t ldo, int64_t n_rows, int64_t n_feat, int reduce, const LnEpilogue& ln, hipStream_t s) {
    constexpr int RPW = gte::kWave / G;
    int passes = 4;
    while (passes > 1 && gte::ceil_div(n_rows, (int64_t)4 * RPW * passes) < (int64_t)16 * gte::device_props().cus) passes /= 2;
    const int rows_per_block = 4 * RPW * passes;
    hipLaunchKernelGGL((spmm_csr_kernel<F32, G, CPL, true, true, MASK>), dim3((unsigned)gte::ceil_div(n_rows, rows_per_block)), dim3(256), 0,
                       s, indptr, indices, ew, x, ldx, out, ldo, (int)n_rows, (int)n_feat, reduce, rows_per_block, ln);
    return gte::check_launch("spmm_csr_accumulate_ln");
}

int spmm_entry(bool accumulate, const int32_t* indptr, const int32_t* indices, const float* eweight,
               const void* x, int64_t ldx, void* out, int64_t ldo, int64_t n_rows, int64_t n_feat,
               int dtype, int reduce, void* stream) {
    if (n_rows < 0 || n_feat < 0 || n_rows > INT32_MAX || n_feat > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr: bad sizes n_rows=%lld n_feat=%lld", (long long)n_rows,
                         (long long)n_feat);
    if (n_rows == 0 || n_feat == 0) return GTE_OK;
    if (!indptr || !x || !out) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr: null pointer");
    if (ldx < n_feat || ldo < n_feat) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr: ld < n_feat");
    if (reduce != GTE_REDUCE_SUM && reduce != GTE_REDUCE_MEAN)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr: reduce must be 0 (sum) or 1 (mean)");
    hipStream_t s = gte::as_stream(stream);
    if (dtype == GTE_F32) {
        return accumulate ? dispatch<F32, true>(indptr, indices, eweight, x, ldx, out, ldo, n_rows, n_feat, reduce, s)
                          : dispatch<F32, false>(indptr, indices, eweight, x, ldx, out, ldo, n_rows, n_feat, reduce, s);
    } else if (dtype == GTE_BF16) {
        return accumulate ? dispatch<BF16, true>(indptr, indices, eweight, x, ldx, out, ldo, n_rows, n_feat, reduce, s)
                          : dispatch<BF16, false>(indptr, indices, eweight, x, ldx, out, ldo, n_rows, n_feat, reduce, s);
    }
    return gte::fail(GTE_ERR_UNSUPPORTED, "spmm_csr: dtype %d", dtype);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Edge-parallel aggregation with a wavefront-level segmented reduction (skewed graphs: hub rows)
// ---------------------------------------------------------------------------------------------------------------
// The row kernels above give a destination row to ONE lane group, which walks the row's edges four at a time: a 3 000-edge hub
// is 750 dependent rounds on one wave while the rest of the chip idles.  Here the unit of work is a SEGMENT of 64 consecutive
// edges of the CSR order, one wave per segment whatever rows it cuts: the wave reads its 64 (source, weight) pairs coalesced,
// one per lane, and reduces them SEGMENTED BY DESTINATION ROW -- lanes across the feature row (16-byte chunks), edges
// broadcast from their lanes, four source rows in flight, a row's sum closed when its last edge (or the segment's) is reached.
// Rows that lie inside one segment are written directly; a row cut by a segment boundary leaves partial sums -- TAIL (the row
// starts in the segment and runs on) or HEAD (it started earlier) -- in a carry buffer, and a second pass adds a row's TAIL and
// HEADs in segment order.  Fixed segmentation, fixed order: bit-reproducible run to run; rows inside a segment are summed in CSR
// order exactly as the row kernels do.  Rows without edges are zero-filled by the segment holding the last edge before them.
constexpr int EP_SEG = 64;

template <int CPL>
__global__ void __launch_bounds__(256)
spmm_edge_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ ew,
                 const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo, int n_rows, int n_edges, int n_feat,
                 int reduce, float* __restrict__ carry, int32_t* __restrict__ carry_row, int64_t ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = blockIdx.x * 4 + wave;
    const int e0 = seg * EP_SEG;
    if (e0 >= n_edges) return;
    const int e1 = min(e0 + EP_SEG, n_edges);
    const int my_e = e0 + lane;
    int my_u = 0;
    float my_w = 0.f;
    if (my_e < e1) { my_u = indices[my_e]; my_w = ew ? ew[my_e] : 1.0f; }
    // the row of the segment's first edge: the last r with indptr[r] <= e0 (uniform binary search)
    int lo = 0, hi = n_rows - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (indptr[mid] <= e0) lo = mid; else hi = mid - 1;
    }
    int r = lo;
    int32_t* crow = carry_row + 2 * (int64_t)seg;              // {row of the HEAD partial, row of the TAIL partial} or -1
    if (lane == 0) { crow[0] = -1; crow[1] = -1; }
    float* chead = carry + (int64_t)seg * 2 * ldc;
    float* ctail = chead + ldc;
    const int nchunk = (n_feat + 3) / 4;
    auto zero_rows = [&](int ra, int rb) {                     // rows [ra, rb) have no edges
        for (int rr = ra; rr < rb; ++rr)
            for (int c = lane; c < nchunk; c += 64) {
                float* o = out + (int64_t)rr * ldo + 4 * c;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (4 * c + q < n_feat) o[q] = 0.f;
            }
    };
    if (seg == 0 && r > 0) zero_rows(0, r);                    // (leading empty rows)
    int e = e0;
    while (e < e1) {
        const int row_lo = indptr[r], row_hi = indptr[r + 1];
        const int se = min(row_hi, e1);
        const bool starts = row_lo >= e0, ends = row_hi <= e1;
        const float scale = (reduce == GTE_REDUCE_MEAN) ? 1.0f / (float)(row_hi - row_lo) : 1.0f;
        for (int cb = 0; cb < nchunk; cb += 64 * CPL) {
            float acc[CPL][4];
#pragma unroll
            for (int j = 0; j < CPL; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
            for (int t = e; t < se; t += kEdgeUnroll) {
                int u[kEdgeUnroll];
                float w[kEdgeUnroll];
#pragma unroll
                for (int k = 0; k < kEdgeUnroll; ++k) {
                    const int tt = min(t + k, se - 1) - e0;      // past the end: the last valid source (cache hit), weight 0
                    u[k] = __shfl(my_u, tt, 64);
                    w[k] = (t + k < se) ? __shfl(my_w, tt, 64) : 0.f;
                }
                float v[kEdgeUnroll][CPL][4];
#pragma unroll
                for (int k = 0; k < kEdgeUnroll; ++k) {
                    const float* xr = x + (int64_t)u[k] * ldx;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int c = cb + lane + 64 * j;
                        if (4 * c + 3 < n_feat) {
                            const f4u tq = *reinterpret_cast<const f4u*>(xr + 4 * c);
                            v[k][j][0] = tq.x; v[k][j][1] = tq.y; v[k][j][2] = tq.z; v[k][j][3] = tq.w;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[k][j][q] = 4 * c + q < n_feat ? xr[4 * c + q] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < kEdgeUnroll; ++k)
#pragma unroll
                    for (int j = 0; j < CPL; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[j][q] = fmaf(w[k], v[k][j][q], acc[j][q]);
            }
            // the row's sum over this segment: final (scaled, to out) or a partial (unscaled, to the carry buffer)
            float* dst = (starts && ends) ? out + (int64_t)r * ldo : (starts ? ctail : chead);
            const float sc = (starts && ends) ? scale : 1.0f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int c = cb + lane + 64 * j;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (4 * c + q < n_feat) dst[4 * c + q] = acc[j][q] * sc;
            }
        }
        if (lane == 0 && !(starts && ends)) crow[starts ? 1 : 0] = r;
        e = se;
        if (ends) {                                            // on to the next row that has edges; empty rows in between are zero
            int rn = r + 1;
            while (rn < n_rows && indptr[rn + 1] == row_hi) ++rn;
            if (rn > r + 1) zero_rows(r + 1, rn);
            r = rn;
        }
    }
}

// second pass: the rows cut by segment boundaries.  One wave per segment that holds a TAIL: the row's TAIL and the HEADs of the
// following segments (their number follows from the row's end), added in segment order.
__global__ void __launch_bounds__(256)
spmm_edge_carry_kernel(const int32_t* __restrict__ indptr, const float* __restrict__ carry, const int32_t* __restrict__ carry_row,
                       int64_t ldc, float* __restrict__ out, int64_t ldo, int n_seg, int n_feat, int reduce) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = blockIdx.x * 4 + wave;
    if (seg >= n_seg) return;
    const int r = carry_row[2 * (int64_t)seg + 1];
    if (r < 0) return;
    const int row_lo = indptr[r], row_hi = indptr[r + 1];
    const int last = (row_hi - 1) / EP_SEG;                    // the segment of the row's last edge
    const float scale = (reduce == GTE_REDUCE_MEAN) ? 1.0f / (float)(row_hi - row_lo) : 1.0f;
    for (int c = lane; 4 * c < n_feat; c += 64) {
        float s[4];
        const float* t = carry + ((int64_t)seg * 2 + 1) * ldc + 4 * c;
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = 4 * c + q < n_feat ? t[q] : 0.f;
        for (int g = seg + 1; g <= last; ++g) {
            const float* h = carry + (int64_t)g * 2 * ldc + 4 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += 4 * c + q < n_feat ? h[q] : 0.f;
        }
        float* o = out + (int64_t)r * ldo + 4 * c;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (4 * c + q < n_feat) o[q] = s[q] * scale;
    }
}

extern "C" int64_t gte_spmm_csr_edge_workspace_bytes(int64_t n_edges, int64_t n_feat) {
    const int64_t n_seg = gte::ceil_div(n_edges > 0 ? n_edges : 1, EP_SEG), ldc = gte::round_up(n_feat > 0 ? n_feat : 1, 4);
    return gte::round_up(n_seg * 2 * ldc * 4 + n_seg * 2 * 4, 256);
}

// out[v, :] = scale_v * sum_e w[e] x[indices[e], :] with the work split by EDGES (64-edge segments, one wave each) instead of by
// rows: the aggregation for graphs with hub rows (max in-degree > 64).  fp32; results equal gte_spmm_csr's up to the summation
// order of rows that span segments.  workspace: gte_spmm_csr_edge_workspace_bytes(n_edges, n_feat).
extern "C" int gte_spmm_csr_edge(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* x, int64_t ldx,
                                 float* out, int64_t ldo, int64_t n_rows, int64_t n_edges, int64_t n_feat, int reduce, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    if (n_rows < 0 || n_edges < 0 || n_feat < 0 || n_rows > INT32_MAX || n_edges > INT32_MAX - EP_SEG || n_feat > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_edge: bad sizes");
    if (n_rows == 0 || n_feat == 0) return GTE_OK;
    if (!indptr || !x || !out || (n_edges > 0 && (!indices || !workspace))) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_edge: null pointer");
    if (ldx < n_feat || ldo < n_feat) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_edge: ld < n_feat");
    if (reduce != GTE_REDUCE_SUM && reduce != GTE_REDUCE_MEAN) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_edge: reduce must be 0 or 1");
    hipStream_t s = gte::as_stream(stream);
    if (n_edges == 0) {                                        // no edges: every row is zero
        if (hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)n_feat * 4, (size_t)n_rows, s) != hipSuccess)
            return gte::fail(GTE_ERR_LAUNCH, "spmm_csr_edge: memset failed");
        return GTE_OK;
    }
    if (workspace_bytes < gte_spmm_csr_edge_workspace_bytes(n_edges, n_feat))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "spmm_csr_edge: workspace too small");
    const int64_t n_seg = gte::ceil_div(n_edges, EP_SEG), ldc = gte::round_up(n_feat, 4);
    float* carry = reinterpret_cast<float*>(workspace);
    int32_t* crow = reinterpret_cast<int32_t*>(carry + n_seg * 2 * ldc);
    const dim3 grid((unsigned)gte::ceil_div(n_seg, 4)), block(256);
    const int64_t nchunk = gte::ceil_div(n_feat, 4);
#define GTE_EP(CPL)                                                                                                          \
    hipLaunchKernelGGL((spmm_edge_kernel<CPL>), grid, block, 0, s, indptr, indices, eweight, x, ldx, out, ldo, (int)n_rows, \
                       (int)n_edges, (int)n_feat, reduce, carry, crow, ldc)
    if (nchunk <= 64) GTE_EP(1); else if (nchunk <= 128) GTE_EP(2); else GTE_EP(4);
#undef GTE_EP
    hipLaunchKernelGGL(spmm_edge_carry_kernel, grid, block, 0, s, indptr, carry, crow, ldc, out, ldo, (int)n_seg, (int)n_feat, reduce);
    return gte::check_launch("spmm_csr_edge");
}

extern "C" int gte_spmm_csr(const int32_t* indptr, const int32_t* indices, const float* eweight, const void* x,
                            int64_t ldx, void* out, int64_t ldo, int64_t n_rows, int64_t n_feat, int dtype,
                            int reduce, void* stream) {
    return spmm_entry(false, indptr, indices, eweight, x, ldx, out, ldo, n_rows, n_feat, dtype, reduce, stream);
}

extern "C" int gte_spmm_csr_accumulate(const int32_t* indptr, const int32_t* indices, const float* eweight,
                                       const void* x, int64_t ldx, void* out, int64_t ldo, int64_t n_rows,
                                       int64_t n_feat, int dtype, int reduce, void* stream) {
    return spmm_entry(true, indptr, indices, eweight, x, ldx, out, ldo, n_rows, n_feat, dtype, reduce, stream);
}

// out (P3 image, csrc/p3.h) = scale_v * sum w x[u]: the aggregation whose result only feeds planes GEMMs (q = A_w^T (norm dz), the
// aggregated input of an aggregate-first layer).  Any n_feat: the image columns from n_feat up to the next multiple of 16 are
// written as zeros.
extern "C" int gte_spmm_csr_p3(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* x, int64_t ldx,
                               void* outp3, int64_t ldp, int64_t n_rows, int64_t n_feat, int reduce, void* stream) {
    if (n_rows < 0 || n_feat <= 0 || n_rows > INT32_MAX || n_feat > INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_p3: bad sizes");
    if (n_rows == 0) return GTE_OK;
    if (!indptr || !x || !outp3) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_p3: null pointer");
    if (ldx < n_feat || ldp < p3::row_bytes(n_feat) || ldp % 16 != 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_p3: leading dimension too small");
    if (reduce != GTE_REDUCE_SUM && reduce != GTE_REDUCE_MEAN) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_p3: reduce must be 0 or 1");
    // `out` of the fp32 path is unused when the image is written; pass x (never dereferenced for stores)
    return dispatch<F32, false>(indptr, indices, eweight, x, ldx, const_cast<float*>(x), ldx, n_rows, n_feat, reduce,
                                gte::as_stream(stream), reinterpret_cast<char*>(outp3), ldp);
}

// Widths the fused aggregation + LayerNorm kernel covers: any n_feat up to 1024 (a row lives in the registers of one lane group).
// n_feat % 4 != 0 (or, with a P3 image of y, n_feat % 16 != 0) needs PADDED rows: x, z and y allocated to the next multiple of
// 4 (16) floats per row (ld >= that); the padding of x is read (zeros), the padding of z / y / the image is written as zeros.
extern "C" int gte_spmm_csr_accumulate_ln_supported(int64_t n_feat) {
    return (n_feat >= 1 && n_feat <= 1024) ? 1 : 0;
}

static int accumulate_ln_impl(const int32_t* indptr, const int32_t* indices, const float* eweight,
                              const float* x, int64_t ldx, float* z, int64_t ldz, int64_t n_rows,
                              int64_t n_feat, int reduce, const float* gamma, const float* beta, float eps,
                              int relu, float* y, int64_t ldy, float* stats, void* yp3, int64_t ldyp3, void* stream) {
    if (n_rows < 0 || n_rows > INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_accumulate_ln: bad n_rows");
    if (!gte_spmm_csr_accumulate_ln_supported(n_feat))
        return gte::fail(GTE_ERR_UNSUPPORTED, "spmm_csr_accumulate_ln: needs 1 <= n_feat <= 1024");
    if (n_rows == 0) return GTE_OK;
    if (!indptr || !x || !z || !gamma || !beta || (!y && !yp3)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_accumulate_ln: null pointer");
    // columns processed: the rows' padded width (a multiple of 4; of 16 when the image is written)
    const int64_t np = yp3 ? gte::round_up(n_feat, 16) : gte::round_up(n_feat, 4);
    if (ldx < np || ldz < np || (y && ldy < np))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_accumulate_ln: ld < n_feat rounded up to %d (padded rows)", yp3 ? 16 : 4);
    if (yp3 && (ldyp3 < p3::row_bytes(n_feat) || ldyp3 % 16 != 0))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_accumulate_ln: the P3 image needs ldp >= 96 ceil(n_feat / 16)");
    if (reduce != GTE_REDUCE_SUM && reduce != GTE_REDUCE_MEAN)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_accumulate_ln: reduce must be 0 (sum) or 1 (mean)");
    const LnEpilogue ln = {gamma, beta, eps, relu, y, ldy, stats, reinterpret_cast<char*>(yp3), ldyp3, (int)n_feat};
    hipStream_t s = gte::as_stream(stream);
    const int64_t nchunk = np / 4;
    const bool mask = np != n_feat;
#define GTE_LN(G, CPL)                                                                                                       \
    return mask ? launch_ln<G, CPL, true>(indptr, indices, eweight, x, ldx, z, ldz, n_rows, np, reduce, ln, s)            \
                : launch_ln<G, CPL, false>(indptr, indices, eweight, x, ldx, z, ldz, n_rows, np, reduce, ln, s)
    if (nchunk <= 4) GTE_LN(4, 1);
    if (nchunk <= 8) GTE_LN(8, 1);
    if (nchunk <= 16) GTE_LN(16, 1);
    if (nchunk <= 32) GTE_LN(32, 1);
    if (nchunk <= 64) GTE_LN(64, 1);
    if (nchunk <= 128) GTE_LN(64, 2);
    GTE_LN(64, 4);
#undef GTE_LN
}

extern "C" int gte_spmm_csr_accumulate_ln(const int32_t* indptr, const int32_t* indices, const float* eweight,
                                          const float* x, int64_t ldx, float* z, int64_t ldz, int64_t n_rows,
                                          int64_t n_feat, int reduce, const float* gamma, const float* beta, float eps,
                                          int relu, float* y, int64_t ldy, float* stats, void* stream) {
    if (!y) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_accumulate_ln: null pointer");
    return accumulate_ln_impl(indptr, indices, eweight, x, ldx, z, ldz, n_rows, n_feat, reduce, gamma, beta, eps, relu, y, ldy, stats,
                              nullptr, 0, stream);
}

// ... with y written as a P3 image (yp3; the operand of the next layer's planes GEMM) and / or as fp32 (y, nullable)
extern "C" int gte_spmm_csr_accumulate_ln_p3(const int32_t* indptr, const int32_t* indices, const float* eweight,
                                             const float* x, int64_t ldx, float* z, int64_t ldz, int64_t n_rows,
                                             int64_t n_feat, int reduce, const float* gamma, const float* beta, float eps,
                                             int relu, float* y, int64_t ldy, void* yp3, int64_t ldyp3, float* stats, void* stream) {
    return accumulate_ln_impl(indptr, indices, eweight, x, ldx, z, ldz, n_rows, n_feat, reduce, gamma, beta, eps, relu, y, ldy, stats,
                              yp3, ldyp3, stream);
}
