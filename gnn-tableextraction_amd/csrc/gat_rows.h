// GAT aggregation kernels in the "head per DPP row" layout (heads <= 4, dim <= 64, dim % VEC == 0): included by gat.hip
// inside its anonymous namespace.
//
// One wave64 per node row.  The wave's four 16-lane DPP rows are the heads; lane 16 h + i plays two roles:
//   feature lane : owns VEC = ceil(dim / 16) CONSECUTIVE features h * dim + VEC * i .. of head h -- a gathered source row is
//                  one 2 VEC / 4 VEC-byte load per lane (the lane-per-feature kernels of gat.hip issue 64 j + lane: four
//                  2-byte loads per edge at heads * dim = 256 in bf16);
//   edge lane    : owns (head h, edge slot i) of the current chunk of 16 edges: the attention logit, its exp and the
//                  softmax backward are computed ONCE per (edge, head) instead of once per feature lane (64x redundant
//                  there), and the reductions over the edges of a head are 16-lane DPP butterflies (no LDS pipe).
// An edge's weight reaches the feature lanes of its head with one DPP row broadcast (v_mov_b32_dpp row_newbcast:e); the
// source node id of edge slot e is a v_readlane.  Rows with more than 16 in-edges take further chunks (online softmax in
// the forward, a second gather pass in the backward).  Summation order over the edges = CSR order, as before.
#pragma once

template <int E>
__device__ __forceinline__ float row_bcast(float v) {      // lane E of every 16-lane row, in all lanes of that row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + E, 0xf, 0xf, true));
}
#define GTE_DPP_MAX(v, ctrl) \
    (v) = fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true)))
__device__ __forceinline__ float row_max16(float v) {
    GTE_DPP_MAX(v, 0xB1); GTE_DPP_MAX(v, 0x4E); GTE_DPP_MAX(v, 0x141); GTE_DPP_MAX(v, 0x140);
    return v;
}
__device__ __forceinline__ float row_sum16(float v) { return gte_group_sum<16>(v); }

template <typename T, int VEC> struct RowVec;
template <int VEC> struct RowVec<float, VEC> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[VEC]) {
        if constexpr (VEC == 4) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
        else if constexpr (VEC == 2) { const float2 v = *reinterpret_cast<const float2*>(p); o[0] = v.x; o[1] = v.y; }
        else o[0] = *p;
    }
    static __device__ __forceinline__ void store(float* p, const float (&o)[VEC]) {
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
        else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
        else *p = o[0];
    }
};
template <int VEC> struct RowVec<unsigned short, VEC> {
    static __device__ __forceinline__ void load(const unsigned short* p, float (&o)[VEC]) {
        if constexpr (VEC == 4) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
            o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        } else if constexpr (VEC == 2) {
            const unsigned v = *reinterpret_cast<const unsigned*>(p);
            o[0] = __uint_as_float(v << 16); o[1] = __uint_as_float(v & 0xffff0000u);
        } else o[0] = __uint_as_float(((unsigned)*p) << 16);
    }
};

// f(E) for the edge slots E = 4 G .. 4 G + 3 of group G (compile-time E: the DPP control and the readlane index are immediates)
template <int G, typename F>
__device__ __forceinline__ void edge_group(F&& f) {
    f(std::integral_constant<int, 4 * G>{}); f(std::integral_constant<int, 4 * G + 1>{});
    f(std::integral_constant<int, 4 * G + 2>{}); f(std::integral_constant<int, 4 * G + 3>{});
}
// all slots below cnt, four at a time (slots past cnt inside a group carry weight 0 and a valid node id)
template <typename F>
__device__ __forceinline__ void for_edges(int cnt, F&& f) {
    edge_group<0>(f);
    if (cnt > 4) edge_group<1>(f);
    if (cnt > 8) edge_group<2>(f);
    if (cnt > 12) edge_group<3>(f);
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
gat_rows_fwd_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, const T* __restrict__ z,
                    int64_t ldzz, const float* __restrict__ el, const float* __restrict__ er, const float* __restrict__ bias,
                    float* __restrict__ out, int64_t ldo, float* __restrict__ smax, float* __restrict__ ssum, int n, int H,
                    int D, const GatEpilogue ep) {
    const int lane = threadIdx.x & 63, h = lane >> 4, i = lane & 15;
    const int v = (int)gte_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const bool hv = h < H, fv = hv && VEC * i < D;
    const int f0 = (hv ? h : 0) * D + (fv ? VEC * i : 0);          // first feature of this lane (a valid one for idle lanes)
    const float erv = hv ? er[(int64_t)v * H + h] : 0.f;
    float m = -INFINITY, l = 0.f, acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
    const int lo = indptr[v], hi = indptr[v + 1];
    for (int c0 = lo; c0 < hi; c0 += 16) {
        const int cnt = min(16, hi - c0);
        const int ue = indices[c0 + min(i, cnt - 1)];                // slots past cnt repeat the last edge (weight 0 below)
        // idle rows (h >= H) carry s = 0 so that nothing in them turns into NaN
        const float s = hv ? (i < cnt ? leaky(el[(int64_t)ue * H + h] + erv) : -INFINITY) : 0.f;
        const float mn = fmaxf(m, row_max16(s));
        const float p = (i < cnt && hv) ? __expf(s - mn) : 0.f;
        const float sc = __expf(m - mn);                             // m = -inf on the first chunk: 0
        l = l * sc + row_sum16(p);
        m = mn;
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] *= sc;
        for_edges(cnt, [&](auto E) {
            constexpr int e = decltype(E)::value;
            const int u = __builtin_amdgcn_readlane(ue, e);
            const float a = row_bcast<e>(p);
            float zv[VEC];
            RowVec<T, VEC>::load(z + (int64_t)u * ldzz + f0, zv);
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] = fmaf(a, zv[c], acc[c]);
        });
    }
    if (hv && i == 0) { smax[(int64_t)v * H + h] = m; ssum[(int64_t)v * H + h] = l; }
    float o[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        o[c] = (hi > lo ? acc[c] / l : 0.f) + ((bias && fv) ? bias[f0 + c] : 0.f);
        if (ep.activation == 1) o[c] = o[c] > 0.f ? o[c] : expm1f(o[c]);
    }
    if (fv) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            if (out) out[(int64_t)v * ldo + f0 + c] = o[c];
            if (ep.out_bf16) {
                __bf16 b = (__bf16)o[c];
                ep.out_bf16[(int64_t)v * ep.ldob + f0 + c] = *reinterpret_cast<unsigned short*>(&b);
            }
        }
    }
    if (ep.out_mean) {                                               // mean over the heads: the four rows' lanes i hold the same d
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            float s = fv ? o[c] : 0.f;
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (h == 0 && VEC * i + c < D)
                ep.out_mean[(int64_t)v * ep.ldom + VEC * i + c] = s / (float)H + (ep.mean_bias ? ep.mean_bias[VEC * i + c] : 0.f);
        }
    }
}

// optional: form the effective dout in the destination-side kernel (all NULL / 0: dout is used as given)
struct GatDoutPrepare { const float* act_out; int64_t ldao; float* dfull; int64_t lddf; int mean_heads; };

// Backward, destination side: ds[e, h] = alpha (d alpha - sum_e' alpha d alpha) leaky'(s),  der[v, h] = sum_e ds[e, h];
// d alpha_e,h = <dout[v, h, :], z[u_e, h, :]>: ONE gather pass when the row has <= 16 in-edges (d alpha stays in the edge lanes).
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
gat_rows_bwd_dst_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, const T* __restrict__ z,
                        int64_t ldzz, const float* __restrict__ el, const float* __restrict__ er,
                        const float* __restrict__ smax, const float* __restrict__ ssum, const float* __restrict__ dout,
                        int64_t lddo, float* __restrict__ ds, float* __restrict__ der, int n, int H, int D,
                        const GatDoutPrepare pp) {
    const int lane = threadIdx.x & 63, h = lane >> 4, i = lane & 15;
    const int v = (int)gte_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const bool hv = h < H, fv = hv && VEC * i < D;
    const int f0 = (hv ? h : 0) * D + (fv ? VEC * i : 0);
    float dv[VEC];
    if (pp.dfull) {
        // the gradient w.r.t. the pre-epilogue aggregate, formed here (gat_dout_prepare_kernel's arithmetic) and left in
        // dfull for the source-side kernel and the column sums: the separate 600 MB pass disappears.  Vector accesses:
        // the lane's VEC features are consecutive in all three arrays.
        float g[VEC], o[VEC];
        RowVec<float, VEC>::load(dout + (int64_t)v * lddo + (pp.mean_heads ? (fv ? VEC * i : 0) : f0), g);
        if (pp.act_out) RowVec<float, VEC>::load(pp.act_out + (int64_t)v * pp.ldao + f0, o);
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            if (pp.mean_heads) g[c] /= (float)H;
            if (pp.act_out) g[c] *= o[c] > 0.f ? 1.f : o[c] + 1.f;
            dv[c] = fv ? g[c] : 0.f;
        }
        if (fv) RowVec<float, VEC>::store(pp.dfull + (int64_t)v * pp.lddf + f0, dv);
    } else {
#pragma unroll
        for (int c = 0; c < VEC; ++c) dv[c] = fv ? dout[(int64_t)v * lddo + f0 + c] : 0.f;
    }
    const float erv = hv ? er[(int64_t)v * H + h] : 0.f;
    const float mv = hv ? smax[(int64_t)v * H + h] : 0.f, lv = hv ? ssum[(int64_t)v * H + h] : 1.f;
    const int lo = indptr[v], hi = indptr[v + 1];
    // one chunk: (s, alpha, d alpha) of edge slot i, head h
    auto chunk = [&](int c0, int cnt, float& s, float& a, float& da) {
        const int ue = indices[c0 + min(i, cnt - 1)];
        s = (i < cnt && hv) ? el[(int64_t)ue * H + h] + erv : 0.f;
        a = (i < cnt && hv) ? __expf(leaky(s) - mv) / lv : 0.f;
        da = 0.f;
        for_edges(cnt, [&](auto E) {
            constexpr int e = decltype(E)::value;
            const int u = __builtin_amdgcn_readlane(ue, e);
            float zv[VEC];
            RowVec<T, VEC>::load(z + (int64_t)u * ldzz + f0, zv);
            float pr = 0.f;
#pragma unroll
            for (int c = 0; c < VEC; ++c) pr = fmaf(dv[c], zv[c], pr);
            const float red = row_sum16(pr);
            da = i == e ? red : da;
        });
    };
    float dersum = 0.f;
    if (hi - lo <= 16) {
        if (hi > lo) {
            const int cnt = hi - lo;
            float s, a, da;
            chunk(lo, cnt, s, a, da);
            const float t = row_sum16(a * da);
            const float g = a * (da - t) * (s > 0.f ? 1.f : kSlope);
            if (i < cnt && hv) ds[(int64_t)(lo + i) * H + h] = g;
            dersum = row_sum16(g);
        }
    } else {
        float t = 0.f;
        for (int c0 = lo; c0 < hi; c0 += 16) {                       // pass 1: t = sum_e alpha d alpha
            float s, a, da;
            chunk(c0, min(16, hi - c0), s, a, da);
            t += row_sum16(a * da);
        }
        for (int c0 = lo; c0 < hi; c0 += 16) {                       // pass 2
            const int cnt = min(16, hi - c0);
            float s, a, da;
            chunk(c0, cnt, s, a, da);
            const float g = a * (da - t) * (s > 0.f ? 1.f : kSlope);
            if (i < cnt && hv) ds[(int64_t)(c0 + i) * H + h] = g;
            dersum += row_sum16(g);
        }
    }
    if (hv && i == 0) der[(int64_t)v * H + h] = dersum;
}

// Backward, source side: dz[u, :] = sum_{out-edges (u -> v)} alpha dout[v, :] + del a_l + der a_r,  del[u, h] = sum ds
template <int VEC>
__global__ void __launch_bounds__(256)
gat_rows_bwd_src_kernel(const int32_t* __restrict__ rindptr, const int32_t* __restrict__ rindices,
                        const int32_t* __restrict__ pos_in, const float* __restrict__ el, const float* __restrict__ er,
                        const float* __restrict__ smax, const float* __restrict__ ssum, const float* __restrict__ dout,
                        int64_t lddo, const float* __restrict__ ds, const float* __restrict__ der,
                        const float* __restrict__ a_l, const float* __restrict__ a_r, float* __restrict__ dz, int64_t lddz,
                        float* __restrict__ del, int n, int H, int D) {
    const int lane = threadIdx.x & 63, h = lane >> 4, i = lane & 15;
    const int u = (int)gte_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (u >= n) return;
    const bool hv = h < H, fv = hv && VEC * i < D;
    const int f0 = (hv ? h : 0) * D + (fv ? VEC * i : 0);
    const float elu = hv ? el[(int64_t)u * H + h] : 0.f;
    float acc[VEC], dl = 0.f;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
    const int lo = rindptr[u], hi = rindptr[u + 1];
    for (int c0 = lo; c0 < hi; c0 += 16) {
        const int cnt = min(16, hi - c0);
        const int ve = rindices[c0 + min(i, cnt - 1)];
        const bool ev = i < cnt && hv;
        const float a = ev ? __expf(leaky(elu + er[(int64_t)ve * H + h]) - smax[(int64_t)ve * H + h]) / ssum[(int64_t)ve * H + h] : 0.f;
        dl += row_sum16(ev ? ds[(int64_t)pos_in[c0 + i] * H + h] : 0.f);
        for_edges(cnt, [&](auto E) {
            constexpr int e = decltype(E)::value;
            const int vv = __builtin_amdgcn_readlane(ve, e);
            const float w = row_bcast<e>(a);
            float dvv[VEC];
            RowVec<float, VEC>::load(dout + (int64_t)vv * lddo + f0, dvv);
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] = fmaf(w, dvv[c], acc[c]);
        });
    }
    if (fv) {
        const float dr = der[(int64_t)u * H + h];
#pragma unroll
        for (int c = 0; c < VEC; ++c) dz[(int64_t)u * lddz + f0 + c] = acc[c] + dl * a_l[f0 + c] + dr * a_r[f0 + c];
    }
    if (hv && i == 0) del[(int64_t)u * H + h] = dl;
}

// usable when every head fits one DPP row and the vector loads are aligned
static inline int gat_rows_vec(int H, int D, int64_t ld_gather, int64_t ld_dout, int elem_bytes) {
    if (H > 4 || D > 64) return 0;
    const int vec = D > 32 ? 4 : (D > 16 ? 2 : 1);
    if (D % vec != 0 || ld_gather % vec != 0 || ld_dout % vec != 0) return 0;
    (void)elem_bytes;
    return vec;
}
