// Multi-head graph attention aggregation on gfx950 (BASELINE.json configs[2]: "4-head GAT bf16").
//
// The reference repository has NO attention layer; the layer follows the standard GAT
// formulation (DGL GATConv semantics), restated for the CPU in oracle/gat_cpu.py -- parity unpinned.
//   z = X W (fp32 MFMA GEMM, sage_linear.hip)            z: [N, H*D]
//   el[v,h] = <a_l[h], z[v,h]>,  er[v,h] = <a_r[h], z[v,h]>
//   e_uv,h = LeakyReLU_0.2(el[u,h] + er[v,h]);  alpha = softmax over the in-edges of v;  out[v,h] = sum alpha z[u,h]
//
// Mapping: one wave64 per destination row, lane l owns features l, l+64, ... (NJ per lane; H*D <= 1024), so a
// gathered source row is read with coalesced 256-B wave-instructions.  The edge softmax is ONLINE (running
// max / sum with rescaling): one pass over the in-edges, nothing per-edge is stored -- the backward recomputes
// alpha from el, er and the saved (max, sum) per (node, head).  Gathered rows may be bf16 (fp32 accumulate):
// gte_gat_scores writes the bf16 copy of z in the same pass that computes el / er.
// Backward = kernel A per destination row (d alpha, softmax and LeakyReLU backward -> per-edge ds and der) +
// kernel B per source row over the out-edge CSR (dz = sum alpha dout[v] + del a_l + der a_r, del) + column
// sums (d a_l, d a_r, d bias) folded deterministically.  No float atomics.  Roofline: HBM (gathers).
#include "gte_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int MAXH = 8;
constexpr float kSlope = 0.2f;

struct GatEpilogue {                  // optional epilogue of the forward aggregation (all NULL / 0: plain out = agg + bias)
    int activation;                   // 1: ELU
    unsigned short* out_bf16; int64_t ldob;        // bf16 copy of the (activated) output
    float* out_mean; int64_t ldom; const float* mean_bias;   // mean over heads [n, D] (+ bias[D])
};

// wave-wide sum on DPP modifiers + four readlanes (gte_common.h): no LDS-pipe instruction.  (The __shfl_xor butterfly is six
// ds_bpermute round trips; the backward takes heads x 2 of these per edge.)
__device__ __forceinline__ float wave_sum(float v) { return gte_group_sum<64>(v); }
__device__ __forceinline__ float leaky(float s) { return s > 0.f ? s : kSlope * s; }
__device__ __forceinline__ float ldz(const float* p) { return *p; }
__device__ __forceinline__ float ldz(const unsigned short* p) { return __uint_as_float(((unsigned)*p) << 16); }

// el / er per (node, head); optional bf16 copy of z
__global__ void __launch_bounds__(256)
gat_scores_kernel(const float* __restrict__ z, int64_t ldzf, const float* __restrict__ a_l, const float* __restrict__ a_r,
                  float* __restrict__ el, float* __restrict__ er, unsigned short* __restrict__ zb, int64_t ldzb, int n,
                  int H, int D) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const float* zr = z + (int64_t)v * ldzf;
    for (int h = 0; h < H; ++h) {
        float sl = 0.f, sr = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float x = zr[h * D + d];
            sl = fmaf(x, a_l[h * D + d], sl);
            sr = fmaf(x, a_r[h * D + d], sr);
            if (zb) { __bf16 b = (__bf16)x; zb[(int64_t)v * ldzb + h * D + d] = *reinterpret_cast<unsigned short*>(&b); }
        }
        sl = wave_sum(sl);
        sr = wave_sum(sr);
        if (lane == 0) { el[(int64_t)v * H + h] = sl; er[(int64_t)v * H + h] = sr; }
    }
}

template <typename T, int NJ>
__global__ void __launch_bounds__(256)
gat_aggregate_fwd_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                         const T* __restrict__ z, int64_t ldzz, const float* __restrict__ el,
                         const float* __restrict__ er, const float* __restrict__ bias, float* __restrict__ out,
                         int64_t ldo, float* __restrict__ smax, float* __restrict__ ssum, int n, int H, int D,
                         const GatEpilogue ep) {
    extern __shared__ float s_mean[];                       // [4 waves][H * D], only with ep.out_mean
    const int lane = threadIdx.x & 63;
    const int v = (int)gte_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const int HD = H * D;
    int hj[NJ];
    float erv[NJ], m[NJ], l[NJ], acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = lane + 64 * j;
        hj[j] = f < HD ? f / D : 0;
        erv[j] = er[(int64_t)v * H + hj[j]];
        m[j] = -INFINITY; l[j] = 0.f; acc[j] = 0.f;
    }
    const int lo = indptr[v], hi = indptr[v + 1];
    for (int e = lo; e < hi; ++e) {
        const int u = indices[e];
        const T* zu = z + (int64_t)u * ldzz;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int f = lane + 64 * j;
            if (f < HD) {
                const float s = leaky(el[(int64_t)u * H + hj[j]] + erv[j]);
                const float mn = fmaxf(m[j], s);
                const float sc = __expf(m[j] - mn), p = __expf(s - mn);      // m = -inf at the first edge: sc = 0
                l[j] = l[j] * sc + p;
                acc[j] = acc[j] * sc + p * ldz(zu + f);
                m[j] = mn;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = lane + 64 * j;
        if (f < HD) {
            float o = (hi > lo ? acc[j] / l[j] : 0.f) + (bias ? bias[f] : 0.f);
            if (ep.activation == 1) o = o > 0.f ? o : expm1f(o);                     // ELU between GAT layers, in the epilogue
            if (out) out[(int64_t)v * ldo + f] = o;
            if (ep.out_bf16) {                                                        // the next layer's bf16 GEMM operand
                __bf16 b = (__bf16)o;
                ep.out_bf16[(int64_t)v * ep.ldob + f] = *reinterpret_cast<unsigned short*>(&b);
            }
            if (ep.out_mean) s_mean[(threadIdx.x >> 6) * HD + f] = o;
            if (f % D == 0) { smax[(int64_t)v * H + hj[j]] = m[j]; ssum[(int64_t)v * H + hj[j]] = l[j]; }
        }
    }
    if (ep.out_mean) {                                     // output layer: mean over the heads (+ its bias), same wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float* sm = s_mean + (threadIdx.x >> 6) * HD;
        for (int d = lane; d < D; d += 64) {
            float s = 0.f;
            for (int h = 0; h < H; ++h) s += sm[h * D + d];
            ep.out_mean[(int64_t)v * ep.ldom + d] = s / (float)H + (ep.mean_bias ? ep.mean_bias[d] : 0.f);
        }
    }
}

// dfull[v, f] = (mean_heads ? dout[v, f % D] / H : dout[v, f]) * (act_out ? ELU'(from the activated output) : 1)
__global__ void __launch_bounds__(256)
gat_dout_prepare_kernel(const float* __restrict__ dout, int64_t lddo, const float* __restrict__ act_out, int64_t ldao,
                        float* __restrict__ dfull, int64_t lddf, int64_t n, int H, int D, int mean_heads) {
    const int HD = H * D;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * HD) return;
    const int64_t v = i / HD;
    const int f = (int)(i % HD);
    float g = mean_heads ? dout[v * lddo + f % D] / (float)H : dout[v * lddo + f];
    if (act_out) { const float o = act_out[v * ldao + f]; g *= o > 0.f ? 1.f : o + 1.f; }      // d ELU = 1 or exp(x) = out + 1
    dfull[v * lddf + f] = g;
}

// Backward A: per destination row v -> ds[e, h] (in-CSR order) and der[v, h]
template <typename T, int NJ>
__global__ void __launch_bounds__(256)
gat_bwd_dst_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, const T* __restrict__ z,
                   int64_t ldzz, const float* __restrict__ el, const float* __restrict__ er,
                   const float* __restrict__ smax, const float* __restrict__ ssum, const float* __restrict__ dout,
                   int64_t lddo, float* __restrict__ ds, float* __restrict__ der, int n, int H, int D) {
    const int lane = threadIdx.x & 63;
    const int v = (int)gte_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const int HD = H * D;
    int hj[NJ];
    float dv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = lane + 64 * j;
        hj[j] = f < HD ? f / D : -1;
        dv[j] = f < HD ? dout[(int64_t)v * lddo + f] : 0.f;
    }
    const int lo = indptr[v], hi = indptr[v + 1];
    float erv[MAXH], mv[MAXH], lv[MAXH], t[MAXH], dersum[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) {
        erv[h] = h < H ? er[(int64_t)v * H + h] : 0.f;
        mv[h] = h < H ? smax[(int64_t)v * H + h] : 0.f;
        lv[h] = h < H ? ssum[(int64_t)v * H + h] : 1.f;
        t[h] = 0.f; dersum[h] = 0.f;
    }
    // d alpha_e,h = <dout[v,h,:], z[u,h,:]> : per-head reduction over the lanes that own that head's features
    auto dalpha = [&](int u, float (&da)[MAXH]) {
        const T* zu = z + (int64_t)u * ldzz;
        float prod[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int f = lane + 64 * j;
            prod[j] = f < HD ? dv[j] * ldz(zu + f) : 0.f;
        }
#pragma unroll
        for (int h = 0; h < MAXH; ++h) {
            if (h < H) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) s += (hj[j] == h) ? prod[j] : 0.f;
                da[h] = wave_sum(s);
            } else da[h] = 0.f;
        }
    };
    for (int e = lo; e < hi; ++e) {                        // pass 1: t[h] = sum_e alpha_e dalpha_e
        const int u = indices[e];
        float da[MAXH];
        dalpha(u, da);
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < H) {
                const float a = __expf(leaky(el[(int64_t)u * H + h] + erv[h]) - mv[h]) / lv[h];
                t[h] = fmaf(a, da[h], t[h]);
            }
    }
    for (int e = lo; e < hi; ++e) {                        // pass 2: ds_e = alpha_e (dalpha_e - t) * leaky'(s_e)
        const int u = indices[e];
        float da[MAXH];
        dalpha(u, da);
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < H) {
                const float s = el[(int64_t)u * H + h] + erv[h];
                const float a = __expf(leaky(s) - mv[h]) / lv[h];
                const float g = a * (da[h] - t[h]) * (s > 0.f ? 1.f : kSlope);
                dersum[h] += g;
                if (lane == h) ds[(int64_t)e * H + h] = g;
            }
    }
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
        if (h < H && lane == h) der[(int64_t)v * H + h] = dersum[h];
}

// Backward B: per source row u over the out-edge CSR -> dz[u, :] and del[u, h]
template <int NJ>
__global__ void __launch_bounds__(256)
gat_bwd_src_kernel(const int32_t* __restrict__ rindptr, const int32_t* __restrict__ rindices,
                   const int32_t* __restrict__ pos_in, const float* __restrict__ el, const float* __restrict__ er,
                   const float* __restrict__ smax, const float* __restrict__ ssum, const float* __restrict__ dout,
                   int64_t lddo, const float* __restrict__ ds, const float* __restrict__ der,
                   const float* __restrict__ a_l, const float* __restrict__ a_r, float* __restrict__ dz, int64_t lddz,
                   float* __restrict__ del, int n, int H, int D) {
    const int lane = threadIdx.x & 63;
    const int u = (int)gte_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (u >= n) return;
    const int HD = H * D;
    int hj[NJ];
    float elu[NJ], acc[NJ], dl[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = lane + 64 * j;
        hj[j] = f < HD ? f / D : 0;
        elu[j] = el[(int64_t)u * H + hj[j]];
        acc[j] = 0.f; dl[j] = 0.f;
    }
    const int lo = rindptr[u], hi = rindptr[u + 1];
    for (int e = lo; e < hi; ++e) {
        const int v = rindices[e];
        const int64_t pin = pos_in[e];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int f = lane + 64 * j;
            if (f < HD) {
                const int h = hj[j];
                const float a = __expf(leaky(elu[j] + er[(int64_t)v * H + h]) - smax[(int64_t)v * H + h]) / ssum[(int64_t)v * H + h];
                acc[j] = fmaf(a, dout[(int64_t)v * lddo + f], acc[j]);
                dl[j] += ds[pin * H + h];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = lane + 64 * j;
        if (f < HD) {
            const int h = hj[j];
            dz[(int64_t)u * lddz + f] = acc[j] + dl[j] * a_l[f] + der[(int64_t)u * H + h] * a_r[f];
            if (f % D == 0) del[(int64_t)u * H + h] = dl[j];
        }
    }
}

// column sums over the nodes: da_l[f] = sum_v del[v,h(f)] z[v,f], da_r[f] = sum_v der[v,h(f)] z[v,f], dbias[f] = sum_v dout[v,f]
constexpr int GC_ROWS = 256;
__global__ void __launch_bounds__(256)
gat_colsum_kernel(const float* __restrict__ z, int64_t ldzf, const float* __restrict__ del, const float* __restrict__ der,
                  const float* __restrict__ dout, int64_t lddo, float* __restrict__ partial, int n, int H, int D) {
    const int HD = H * D;
    const int r0 = blockIdx.x * GC_ROWS, r1 = min(r0 + GC_ROWS, n);
    for (int f = threadIdx.x; f < HD; f += 256) {
        const int h = f / D;
        float a = 0.f, b = 0.f, c = 0.f;
        int v = r0;
        // eight rows' loads in flight per thread (the one-row-at-a-time loop was latency-bound: 115 us at 200 k x 256);
        // the sums keep the row order
        for (; v + 8 <= r1; v += 8) {
            float x[8], dl[8], dr[8], dd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = z[(int64_t)(v + u) * ldzf + f];
                dl[u] = del[(int64_t)(v + u) * H + h];
                dr[u] = der[(int64_t)(v + u) * H + h];
                dd[u] = dout[(int64_t)(v + u) * lddo + f];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a = fmaf(dl[u], x[u], a); b = fmaf(dr[u], x[u], b); c += dd[u]; }
        }
        for (; v < r1; ++v) {
            const float x = z[(int64_t)v * ldzf + f];
            a = fmaf(del[(int64_t)v * H + h], x, a);
            b = fmaf(der[(int64_t)v * H + h], x, b);
            c += dout[(int64_t)v * lddo + f];
        }
        float* pp = partial + (int64_t)blockIdx.x * 3 * HD;
        pp[f] = a; pp[HD + f] = b; pp[2 * HD + f] = c;
    }
}
// fold the block partials in a fixed order: block = 64 columns x 16 slices of the block list, slice sums added in slice order
// through LDS.  (One thread per column walking ALL block partials one load at a time -- 3 125 dependent iterations at 200 k
// nodes -- took 800 us per layer, a quarter of the cfg3 step.)
__global__ void __launch_bounds__(1024)
gat_fold_kernel(const float* __restrict__ partial, int nblk, int HD, float* __restrict__ da_l, float* __restrict__ da_r,
                float* __restrict__ dbias) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane;
    float a = 0.f, b = 0.f, c = 0.f;
    if (f < HD) {
#pragma unroll 4
        for (int k = slice; k < nblk; k += 16) {
            const float* pp = partial + (int64_t)k * 3 * HD;
            a += pp[f]; b += pp[HD + f]; c += pp[2 * HD + f];
        }
    }
    red[0][slice][lane] = a; red[1][slice][lane] = b; red[2][slice][lane] = c;
    __syncthreads();
    if (slice < 3 && f < HD) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[slice][k][lane];
        float* dst = slice == 0 ? da_l : slice == 1 ? da_r : dbias;
        if (dst) dst[f] = s;
    }
}

#include "gat_rows.h"

// GTE_GAT_ROWS=0 keeps the lane-per-feature kernels (A/B measurements)
bool gat_rows_enabled() {                   // read per call: tests switch it inside one process
    return !GTE_MEASURE_OFF("GTE_GAT_ROWS");
}

int check_dims(int64_t n, int H, int D, const char* who) {
    if (n < 0 || n > INT32_MAX || H <= 0 || H > MAXH || D <= 0 || (int64_t)H * D > 1024)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "%s: need 1 <= heads <= 8 and heads*dim <= 1024", who);
    return GTE_OK;
}

}  // namespace

#define GTE_NJ_DISPATCH(HD, CALL)            \
    if ((HD) <= 64) { CALL(1); }             \
    else if ((HD) <= 128) { CALL(2); }       \
    else if ((HD) <= 256) { CALL(4); }       \
    else if ((HD) <= 512) { CALL(8); }       \
    else { CALL(16); }

extern "C" int gte_gat_scores(const float* z, int64_t ldz, const float* a_l, const float* a_r, float* el, float* er,
                              void* z_bf16, int64_t ldzb, int64_t n_nodes, int heads, int dim, void* stream) {
    if (int rc = check_dims(n_nodes, heads, dim, "gat_scores")) return rc;
    if (n_nodes == 0) return GTE_OK;
    if (!z || !a_l || !a_r || !el || !er) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gat_scores: null pointer");
    hipLaunchKernelGGL(gat_scores_kernel, dim3((unsigned)gte::ceil_div(n_nodes, 4)), dim3(256), 0, gte::as_stream(stream), z,
                       ldz, a_l, a_r, el, er, (unsigned short*)z_bf16, ldzb, (int)n_nodes, heads, dim);
    return gte::check_launch("gat_scores");
}

extern "C" int gte_gat_aggregate_fwd_ex(const int32_t* indptr, const int32_t* indices, const void* z, int64_t ldz, int dtype,
                                        const float* el, const float* er, const float* bias, float* out, int64_t ldo,
                                        float* smax, float* ssum, int64_t n_nodes, int heads, int dim, int activation,
                                        void* out_bf16, int64_t ldob, float* out_mean, int64_t ldom, const float* mean_bias,
                                        void* stream) {
    if (int rc = check_dims(n_nodes, heads, dim, "gat_aggregate_fwd")) return rc;
    if (n_nodes == 0) return GTE_OK;
    if (!indptr || !z || !el || !er || (!out && !out_mean) || !smax || !ssum)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gat_aggregate_fwd: null pointer");
    if (activation != 0 && activation != 1) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gat_aggregate_fwd: activation 0 | 1 (ELU)");
    const int HD = heads * dim;
    if ((out_bf16 && ldob < HD) || (out_mean && ldom < dim) || (out && ldo < HD))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gat_aggregate_fwd: leading dimension too small");
    const GatEpilogue ep = {activation, (unsigned short*)out_bf16, ldob, out_mean, ldom, mean_bias};
    const size_t shm = out_mean ? (size_t)4 * HD * sizeof(float) : 0;
    dim3 grid((unsigned)gte::ceil_div(n_nodes, 4)), block(256);
    hipStream_t s = gte::as_stream(stream);
    if (const int vec = gat_rows_enabled() ? gat_rows_vec(heads, dim, ldz, ldz, 0) : 0) {
#define ROWS(VEC)                                                                                                          \
    if (dtype == GTE_BF16)                                                                                                 \
        hipLaunchKernelGGL((gat_rows_fwd_kernel<unsigned short, VEC>), grid, block, 0, s, indptr, indices,                \
                           (const unsigned short*)z, ldz, el, er, bias, out, ldo, smax, ssum, (int)n_nodes, heads, dim, ep); \
    else                                                                                                                   \
        hipLaunchKernelGGL((gat_rows_fwd_kernel<float, VEC>), grid, block, 0, s, indptr, indices, (const float*)z, ldz, el, \
                           er, bias, out, ldo, smax, ssum, (int)n_nodes, heads, dim, ep)
        if (vec == 4) { ROWS(4); } else if (vec == 2) { ROWS(2); } else { ROWS(1); }
#undef ROWS
        return gte::check_launch("gat_aggregate_fwd");
    }
#define CALL(NJ)                                                                                                          \
    if (dtype == GTE_BF16)                                                                                                \
        hipLaunchKernelGGL((gat_aggregate_fwd_kernel<unsigned short, NJ>), grid, block, shm, s, indptr, indices,         \
                           (const unsigned short*)z, ldz, el, er, bias, out, ldo, smax, ssum, (int)n_nodes, heads, dim, ep); \
    else                                                                                                                  \
        hipLaunchKernelGGL((gat_aggregate_fwd_kernel<float, NJ>), grid, block, shm, s, indptr, indices, (const float*)z, \
                           ldz, el, er, bias, out, ldo, smax, ssum, (int)n_nodes, heads, dim, ep)
    GTE_NJ_DISPATCH(HD, CALL)
#undef CALL
    return gte::check_launch("gat_aggregate_fwd");
}

extern "C" int gte_gat_aggregate_fwd(const int32_t* indptr, const int32_t* indices, const void* z, int64_t ldz, int dtype,
                                     const float* el, const float* er, const float* bias, float* out, int64_t ldo,
                                     float* smax, float* ssum, int64_t n_nodes, int heads, int dim, void* stream) {
    return gte_gat_aggregate_fwd_ex(indptr, indices, z, ldz, dtype, el, er, bias, out, ldo, smax, ssum, n_nodes, heads, dim, 0,
                                    nullptr, 0, nullptr, 0, nullptr, stream);
}

extern "C" int gte_gat_dout_prepare(const float* dout, int64_t lddo, const float* act_out, int64_t ldao, float* dfull,
                                    int64_t lddf, int64_t n_nodes, int heads, int dim, int mean_heads, void* stream) {
    if (int rc = check_dims(n_nodes, heads, dim, "gat_dout_prepare")) return rc;
    if (n_nodes == 0) return GTE_OK;
    if (!dout || !dfull) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gat_dout_prepare: null pointer");
    const int64_t work = n_nodes * heads * dim;
    hipLaunchKernelGGL(gat_dout_prepare_kernel, dim3((unsigned)gte::ceil_div(work, 256)), dim3(256), 0, gte::as_stream(stream), dout,
                       lddo, act_out, ldao, dfull, lddf, n_nodes, heads, dim, mean_heads);
    return gte::check_launch("gat_dout_prepare");
}

extern "C" int64_t gte_gat_bwd_workspace_bytes(int64_t n_nodes, int heads, int dim) {
    return gte::round_up(gte::ceil_div(n_nodes > 0 ? n_nodes : 1, GC_ROWS) * 3 * (int64_t)heads * dim * 4, 256);
}

extern "C" int gte_gat_aggregate_bwd(const int32_t* indptr, const int32_t* indices, const int32_t* rindptr,
                                     const int32_t* rindices, const int32_t* pos_in, const void* z, int64_t ldz,
                                     int dtype, const float* z_f32, int64_t ldzf, const float* el, const float* er,
                                     const float* smax, const float* ssum, const float* a_l, const float* a_r,
                                     const float* dout, int64_t lddo, float* ds, float* der, float* del, float* dz,
                                     int64_t lddz, float* da_l, float* da_r, float* dbias, int64_t n_nodes, int heads,
                                     int dim, void* workspace, int64_t workspace_bytes, void* stream) {
    return gte_gat_aggregate_bwd_ex(indptr, indices, rindptr, rindices, pos_in, z, ldz, dtype, z_f32, ldzf, el, er, smax, ssum, a_l,
                                    a_r, dout, lddo, nullptr, 0, 0, nullptr, 0, ds, der, del, dz, lddz, da_l, da_r, dbias, n_nodes,
                                    heads, dim, workspace, workspace_bytes, stream);
}

extern "C" int gte_gat_aggregate_bwd_ex(const int32_t* indptr, const int32_t* indices, const int32_t* rindptr,
                                        const int32_t* rindices, const int32_t* pos_in, const void* z, int64_t ldz,
                                        int dtype, const float* z_f32, int64_t ldzf, const float* el, const float* er,
                                        const float* smax, const float* ssum, const float* a_l, const float* a_r,
                                        const float* dout_in, int64_t lddo_in, const float* act_out, int64_t ldao, int mean_heads,
                                        float* dfull, int64_t lddf, float* ds, float* der, float* del, float* dz,
                                        int64_t lddz, float* da_l, float* da_r, float* dbias, int64_t n_nodes, int heads,
                                        int dim, void* workspace, int64_t workspace_bytes, void* stream) {
    if (int rc = check_dims(n_nodes, heads, dim, "gat_aggregate_bwd")) return rc;
    if (n_nodes == 0) return GTE_OK;
    if (dfull && lddf < (int64_t)heads * dim) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gat_aggregate_bwd: lddf too small");
    // effective dout: dfull when given (formed by the destination-side kernel, or by the prepare kernel where that kernel
    // is not the row-layout one), else dout_in
    const float* dout = dfull ? dfull : dout_in;
    const int64_t lddo = dfull ? lddf : lddo_in;
    if (!indptr || !rindptr || !z || !z_f32 || !el || !er || !smax || !ssum || !a_l || !a_r || !dout_in || !ds || !der ||
        !del || !dz || !workspace)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gat_aggregate_bwd: null pointer");
    if (workspace_bytes < gte_gat_bwd_workspace_bytes(n_nodes, heads, dim))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "gat_aggregate_bwd: workspace too small");
    const int HD = heads * dim;
    dim3 grid((unsigned)gte::ceil_div(n_nodes, 4)), block(256);
    hipStream_t s = gte::as_stream(stream);
    const int vec = gat_rows_enabled() ? gat_rows_vec(heads, dim, ldz, lddo, 0) : 0;
    GatDoutPrepare pp = {act_out, ldao, dfull, lddf, mean_heads};
    const bool fuse_prepare = vec && dfull && lddo_in % vec == 0 && (!act_out || ldao % vec == 0) && lddf % vec == 0;
    if (dfull && !fuse_prepare) {                          // the prepare pass as its own launch (lane-per-feature kernels, odd strides)
        const int64_t work = n_nodes * heads * dim;
        hipLaunchKernelGGL(gat_dout_prepare_kernel, dim3((unsigned)gte::ceil_div(work, 256)), dim3(256), 0, s, dout_in, lddo_in,
                           act_out, ldao, dfull, lddf, n_nodes, heads, dim, mean_heads);
        pp.dfull = nullptr;
    }
    const float* dout_dst = fuse_prepare ? dout_in : dout;   // what the destination-side kernel reads
    const int64_t lddo_dst = fuse_prepare ? lddo_in : lddo;
    if (vec) {
#define ROWS(VEC)                                                                                                          \
    if (dtype == GTE_BF16)                                                                                                 \
        hipLaunchKernelGGL((gat_rows_bwd_dst_kernel<unsigned short, VEC>), grid, block, 0, s, indptr, indices,            \
                           (const unsigned short*)z, ldz, el, er, smax, ssum, dout_dst, lddo_dst, ds, der, (int)n_nodes, heads, \
                           dim, pp);                                                                                       \
    else                                                                                                                   \
        hipLaunchKernelGGL((gat_rows_bwd_dst_kernel<float, VEC>), grid, block, 0, s, indptr, indices, (const float*)z, ldz, \
                           el, er, smax, ssum, dout_dst, lddo_dst, ds, der, (int)n_nodes, heads, dim, pp);                 \
    hipLaunchKernelGGL((gat_rows_bwd_src_kernel<VEC>), grid, block, 0, s, rindptr, rindices, pos_in, el, er, smax, ssum,  \
                       dout, lddo, ds, der, a_l, a_r, dz, lddz, del, (int)n_nodes, heads, dim)
        if (vec == 4) { ROWS(4); } else if (vec == 2) { ROWS(2); } else { ROWS(1); }
#undef ROWS
    } else {
#define CALL(NJ)                                                                                                           \
    if (dtype == GTE_BF16)                                                                                                 \
        hipLaunchKernelGGL((gat_bwd_dst_kernel<unsigned short, NJ>), grid, block, 0, s, indptr, indices,                  \
                           (const unsigned short*)z, ldz, el, er, smax, ssum, dout, lddo, ds, der, (int)n_nodes, heads, dim); \
    else                                                                                                                   \
        hipLaunchKernelGGL((gat_bwd_dst_kernel<float, NJ>), grid, block, 0, s, indptr, indices, (const float*)z, ldz, el, \
                           er, smax, ssum, dout, lddo, ds, der, (int)n_nodes, heads, dim);                                 \
    hipLaunchKernelGGL((gat_bwd_src_kernel<NJ>), grid, block, 0, s, rindptr, rindices, pos_in, el, er, smax, ssum, dout,  \
                       lddo, ds, der, a_l, a_r, dz, lddz, del, (int)n_nodes, heads, dim)
    GTE_NJ_DISPATCH(HD, CALL)
#undef CALL
    }
    const int nblk = (int)gte::ceil_div(n_nodes, GC_ROWS);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(gat_colsum_kernel, dim3((unsigned)nblk), dim3(256), 0, s, z_f32, ldzf, del, der, dout, lddo, part,
                       (int)n_nodes, heads, dim);
    hipLaunchKernelGGL(gat_fold_kernel, dim3((unsigned)gte::ceil_div(HD, 64)), dim3(1024), 0, s, part, nblk, HD, da_l, da_r,
                       dbias);
    return gte::check_launch("gat_aggregate_bwd");
}
