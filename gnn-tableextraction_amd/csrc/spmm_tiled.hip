// LDS-staged neighbour aggregation: the same contraction as spmm_csr.hip,
//   out[v, :] = scale_v * sum_{e in row v} w[e] * x[indices[e], :]
// (reference call site src/components/graphs/models.py:53-54, DGL gSpMM), for graphs with locality.
//
// Page graphs are k-NN graphs over words in reading order: consecutive destination rows share most of
// their sources (measured: 32 consecutive rows x 12 in-edges reference ~84 distinct sources, 64 rows
// ~136).  The plain row-per-wave kernel re-fetches every shared source row through L2 once per edge
// (cfg4: 24.6 GB of gathered traffic for 4.2 GB of algorithmic bytes).  Here a workgroup owns a TILE of
// R = 32 destination rows and
//   1. keeps the tile's edge list (local source index + weight) in LDS,
//   2. per 32-float feature chunk stages the tile's DISTINCT source rows into LDS once
//      (global_load_dwordx4, 8 rows x 128 B per wave-instruction; a lane's loads are issued back to back
//      one chunk AHEAD, into registers, so they land under the previous chunk's reduction),
//   3. reduces every destination row from LDS (ds_read_b128, an 8-lane group reads one 128-B row) in CSR
//      order -- same summation order as the plain kernel, bit-identical result,
//   4. writes the 128-B output segments.
// 20 KB of LDS per workgroup -> 8 workgroups (32 waves) per CU; occupancy, not traffic, is what the
// chunk width trades against (see FC below).  HBM-side traffic measured with rocprofv3 on cfg4:
// 2.38 GB fetched + 2.0 GB written = 4.4 GB for 4.2 GB of algorithmic bytes (plain kernel: 10.7 GB).
// The distinct-source lists are graph structure, built once per (batched) graph next to the CSR
// (tile_ptr / tile_src / local index per edge) and reused by every layer, forward and backward, and
// every epoch.  Tiles that do not fit the LDS budget (more than UMAX distinct sources or EMAX edges:
// hub rows) take a direct-gather path inside the same kernel, so any graph is accepted.
// Roofline: HBM; algorithmic bytes per destination row = 2*F*4 + 8*deg + 4.
#include "gte_common.h"

namespace {

constexpr int TILE_R = 32;        // destination rows per tile  (must match the host-side plan)
constexpr int UMAX = 128;         // distinct source rows staged per tile
constexpr int EMAX = 448;         // edges of a tile kept in LDS (32 rows x up to 14 in-edges): 8-byte {weight, local index} entries
#ifndef GTE_TILED_FC
#define GTE_TILED_FC 32
#endif
constexpr int FC = GTE_TILED_FC;  // floats per feature chunk: 32 = one 128-B line per row and 8 workgroups per CU
                                  // (measured on cfg4: FC 32 -> 3.97 TB/s, 64 -> 3.2-3.4, 128 -> 2.4; double-buffered LDS 3.0;
                                  //  64-row tiles (136 distinct sources instead of 2 x 84, 5 workgroups per CU) -> 3.55;
                                  //  {weight, local index} packed into one 8-byte LDS entry: 3.90 -> 4.08)
constexpr int LPR = FC / 4;       // lanes per staged row (16-byte pieces)
constexpr int GPW = 64 / LPR;     // row groups per wave

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

__device__ __forceinline__ float4 ld4_guard(const float* p, int nvalid) {
    // nvalid in 0..4 valid floats at p (feature tail when F % 4 != 0)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4) { const f4u t = *reinterpret_cast<const f4u*>(p); v = make_float4(t.x, t.y, t.z, t.w); }
    else if (nvalid > 0) { v.x = p[0]; if (nvalid > 1) v.y = p[1]; if (nvalid > 2) v.z = p[2]; }
    return v;
}

__device__ __forceinline__ void st4_guard(float* p, float4 v, int nvalid) {
    if (nvalid >= 4) { f4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *reinterpret_cast<f4u*>(p) = t; }
    else if (nvalid > 0) { p[0] = v.x; if (nvalid > 1) p[1] = v.y; if (nvalid > 2) p[2] = v.z; }
}

template <bool ACCUM>
__global__ void __launch_bounds__(256)
spmm_tiled_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                  const uint16_t* __restrict__ lidx, const float* __restrict__ ew,
                  const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ tile_src,
                  const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
                  int n_rows, int n_feat, int reduce) {
#ifdef GTE_TILED_DB
    __shared__ __attribute__((aligned(16))) float s_rows_all[2 * UMAX * FC];   // two chunks: one barrier per chunk
#else
    __shared__ __attribute__((aligned(16))) float s_rows_all[UMAX * FC];       // staged source rows (one chunk)
#endif
    struct Edge { float w; int li; };                          // one ds_read_b64 per edge (two reads with separate arrays)
    __shared__ __attribute__((aligned(8))) Edge s_e[EMAX];
    __shared__ int s_usrc[UMAX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LPR, l16 = lane % LPR;               // GPW row-groups of LPR lanes per wave
    const int ntiles = (n_rows + TILE_R - 1) / TILE_R;
    const int tile = (int)gte_xcd_remap(blockIdx.x, (unsigned)ntiles);
    const int row0 = tile * TILE_R;
    const int row_end = min(row0 + TILE_R, n_rows);
    const int e0 = indptr[row0], e1 = indptr[row_end];
    const int u0 = tile_ptr[tile], nu = tile_ptr[tile + 1] - u0;
    const int ne = e1 - e0;
    const bool staged = (nu <= UMAX) && (ne <= EMAX);          // block-uniform

    if (staged) {
        for (int i = tid; i < ne; i += 256) { s_e[i].li = lidx[e0 + i]; s_e[i].w = ew ? ew[e0 + i] : 1.0f; }
        for (int i = tid; i < nu; i += 256) s_usrc[i] = tile_src[u0 + i];
    }
    __syncthreads();

    // Register-pipelined chunks: while chunk c is reduced from LDS, the distinct source rows of chunk c+1
    // are already in flight into registers (8 x 16 B per lane = all UMAX rows of the tile at once).
    constexpr int NLD = UMAX / (4 * GPW);                            // loads per lane per chunk
    float4 pre[NLD];
    auto issue_chunk = [&](int c0) {
        const int col = c0 + l16 * 4;
        const int nvalid = min(max(n_feat - col, 0), 4);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int ur = wave * GPW + g + 4 * GPW * k;
            const int srow = s_usrc[min(ur, max(nu - 1, 0))];
            pre[k] = (ur < nu) ? ld4_guard(x + (int64_t)srow * ldx + col, nvalid) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (staged && n_feat > 0) issue_chunk(0);

    for (int c0 = 0; c0 < n_feat; c0 += FC) {
        const int col = c0 + l16 * 4;
        const int nvalid = min(max(n_feat - col, 0), 4);
#ifdef GTE_TILED_DB
        float* s_rows = s_rows_all + ((c0 / FC) & 1) * (UMAX * FC);
#else
        float* s_rows = s_rows_all;
#endif
        if (staged) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int ur = wave * GPW + g + 4 * GPW * k;
                if (ur < nu) *reinterpret_cast<float4*>(&s_rows[ur * FC + l16 * 4]) = pre[k];
            }
            __syncthreads();
            if (c0 + FC < n_feat) issue_chunk(c0 + FC);          // lands under this chunk's reduction
        }
        // ---- reduce: wave w owns rows row0 + 8w .. +7, four rows at a time (one per 16-lane group)
#pragma unroll
        for (int pass = 0; pass < TILE_R / (4 * GPW); ++pass) {
            const int r = row0 + wave * (TILE_R / 4) + pass * GPW + g;
            if (r < row_end) {
                const int lo = indptr[r], hi = indptr[r + 1];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (staged) {
                    int e = lo - e0;
                    const int eh = hi - e0;
                    for (; e + 1 < eh; e += 2) {             // two LDS rows in flight
                        const Edge ea = s_e[e], eb = s_e[e + 1];
                        const float w0 = ea.w, w1 = eb.w;
                        const float4 a = *reinterpret_cast<const float4*>(&s_rows[ea.li * FC + l16 * 4]);
                        const float4 b = *reinterpret_cast<const float4*>(&s_rows[eb.li * FC + l16 * 4]);
                        acc.x = fmaf(w0, a.x, acc.x); acc.y = fmaf(w0, a.y, acc.y);
                        acc.z = fmaf(w0, a.z, acc.z); acc.w = fmaf(w0, a.w, acc.w);
                        acc.x = fmaf(w1, b.x, acc.x); acc.y = fmaf(w1, b.y, acc.y);
                        acc.z = fmaf(w1, b.z, acc.z); acc.w = fmaf(w1, b.w, acc.w);
                    }
                    if (e < eh) {
                        const Edge ea = s_e[e];
                        const float w0 = ea.w;
                        const float4 a = *reinterpret_cast<const float4*>(&s_rows[ea.li * FC + l16 * 4]);
                        acc.x = fmaf(w0, a.x, acc.x); acc.y = fmaf(w0, a.y, acc.y);
                        acc.z = fmaf(w0, a.z, acc.z); acc.w = fmaf(w0, a.w, acc.w);
                    }
                } else {                                     // hub tile: gather straight from global memory
                    for (int e = lo; e < hi; ++e) {
                        const float w0 = ew ? ew[e] : 1.0f;
                        const float4 a = ld4_guard(x + (int64_t)indices[e] * ldx + col, nvalid);
                        acc.x = fmaf(w0, a.x, acc.x); acc.y = fmaf(w0, a.y, acc.y);
                        acc.z = fmaf(w0, a.z, acc.z); acc.w = fmaf(w0, a.w, acc.w);
                    }
                }
                const float scale = (reduce == GTE_REDUCE_MEAN) ? (hi > lo ? 1.0f / (float)(hi - lo) : 0.0f) : 1.0f;
                acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
                float* op = out + (int64_t)r * ldo + col;
                if constexpr (ACCUM) {
                    const float4 o = ld4_guard(op, nvalid);
                    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                }
                st4_guard(op, acc, nvalid);
            }
        }
#ifndef GTE_TILED_DB
        if (staged) __syncthreads();                          // s_rows is overwritten by the next chunk
#endif
    }
}

}  // namespace

extern "C" int gte_spmm_tile_rows(void) { return TILE_R; }

extern "C" int gte_spmm_csr_tiled(const int32_t* indptr, const int32_t* indices, const uint16_t* local_index,
                                  const float* eweight, const int32_t* tile_ptr, const int32_t* tile_src,
                                  const float* x, int64_t ldx, float* out, int64_t ldo, int64_t n_rows,
                                  int64_t n_feat, int reduce, int accumulate, void* stream) {
    if (n_rows < 0 || n_feat < 0 || n_rows > INT32_MAX || n_feat > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_tiled: bad sizes");
    if (n_rows == 0 || n_feat == 0) return GTE_OK;
    if (!indptr || !indices || !local_index || !tile_ptr || !tile_src || !x || !out)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_tiled: null pointer");
    if (ldx < n_feat || ldo < n_feat) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_tiled: ld < n_feat");
    if (reduce != GTE_REDUCE_SUM && reduce != GTE_REDUCE_MEAN)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "spmm_csr_tiled: reduce must be 0 (sum) or 1 (mean)");
    const int64_t ntiles = gte::ceil_div(n_rows, TILE_R);
    dim3 grid((unsigned)ntiles), block(256);
    hipStream_t s = gte::as_stream(stream);
    if (accumulate)
        hipLaunchKernelGGL(spmm_tiled_kernel<true>, grid, block, 0, s, indptr, indices, local_index, eweight, tile_ptr,
                           tile_src, x, ldx, out, ldo, (int)n_rows, (int)n_feat, reduce);
    else
        hipLaunchKernelGGL(spmm_tiled_kernel<false>, grid, block, 0, s, indptr, indices, local_index, eweight, tile_ptr,
                           tile_src, x, ldx, out, ldo, (int)n_rows, (int)n_feat, reduce);
    return gte::check_launch("spmm_csr_tiled");
}
