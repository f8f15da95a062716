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
// Two kernels share this structure:
//   spmm_tiled_full_kernel  widths that are a multiple of FC with >= 4 chunks: TWO chunks of distinct source rows in flight
//                           in two register sets (72 VGPRs, 7 workgroups per CU); cfg4: 0.90 ms per pass, 4.6-4.8 TB/s of
//                           algorithmic bytes -- the rate of a device copy of the same matrix (profiles/r01/stream_bw.txt)
//   spmm_tiled_kernel       any width (guarded 16-byte pieces, one chunk in flight, 8 workgroups per CU); cfg4: 1.05 ms
// 20 KB of LDS per workgroup; occupancy, not traffic, is what the chunk width trades against (see FC below).  HBM-side
// traffic measured with rocprofv3 on cfg4: 4.6 GB per launch for 4.2 GB of algorithmic bytes (plain kernel: 10.9 GB).
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
#ifndef GTE_TILED_EMAX_FULL
#define GTE_TILED_EMAX_FULL 512
#endif
constexpr int EMAX_FULL = GTE_TILED_EMAX_FULL;   // the same in spmm_tiled_full_kernel (7 workgroups per CU by registers, so the
                                  // half KB is free there).  cfg4's out-edge CSR has 13 tiles of 449 ... 460 edges: on the direct-gather path
                                  // they were the tail of the launch -- 966 -> 880 us (0.54 -> 0.60 of HBM peak) with 512
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
typedef float v4f __attribute__((ext_vector_type(4)));

// 16-byte store the compiler's s_waitcnt pass does not see.  gfx9 counts loads and stores in ONE vmcnt; with both kinds
// pending the pass treats the counter as unordered and drains it (vmcnt(0)) before the first use of any loaded register,
// which would serialise the two prefetch sets of spmm_tiled_full_kernel.  With the stores hidden the pass sees loads only
// and counts them itself ("at most NLD newer loads outstanding" releases the older set); a store it did not count only
// makes such a wait stricter (loads retire in order among loads).  The loads stay ordinary C++ loads: the compiler knows
// which registers are pending, so any copy it makes of them is preceded by its own wait.  The s_nop covers the hazard the
// compiler would otherwise pad itself: a VALU write to the data registers of a >8-byte store right behind it.
__device__ __forceinline__ void st4_hidden(float* p, float4 v) {
    const v4f t = {v.x, v.y, v.z, v.w};
    // nt: the output rows are written once and not read again by this kernel -- streaming stores leave L2 to the source rows
    // that neighbouring tiles re-read (cfg4, same process, interleaved rounds: 841 -> 821 us; as a pure copy, self-loop graph:
    // 5.37 -> 5.61 TB/s; bitwise identical output).  nt LOADS of the source rows lose that reuse: 841 -> 1205 us.
#ifdef GTE_TILED_PLAIN_STORE
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#endif
}

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
    __shared__ __attribute__((aligned(16))) float s_rows_all[UMAX * FC];       // staged source rows (one chunk)
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

    // Register-pipelined chunks: while chunk c is reduced from LDS, the distinct source rows of chunk c+1 are already in
    // flight into registers (4 x 16 B per lane per chunk = all UMAX rows of the tile at once).
    constexpr int NLD = UMAX / (4 * GPW);                            // loads per lane per chunk
    auto issue_chunk = [&](int c0, float4 (&pre)[NLD]) {
        const int col = c0 + l16 * 4;
        const int nvalid = min(max(n_feat - col, 0), 4);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int ur = wave * GPW + g + 4 * GPW * k;
            const int srow = s_usrc[min(ur, max(nu - 1, 0))];
            pre[k] = (ur < nu) ? ld4_guard(x + (int64_t)srow * ldx + col, nvalid) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage_chunk = [&](float* s_rows, const float4 (&pre)[NLD]) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int ur = wave * GPW + g + 4 * GPW * k;
            if (ur < nu) *reinterpret_cast<float4*>(&s_rows[ur * FC + l16 * 4]) = pre[k];
        }
    };
    // ---- reduce: wave w owns rows row0 + 8w .. +7, one per 8-lane group
    auto reduce_chunk = [&](int c0, const float* s_rows) {
        const int col = c0 + l16 * 4;
        const int nvalid = min(max(n_feat - col, 0), 4);
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
    };

    float* const s_rows = s_rows_all;
    float4 pre[NLD];
    if (staged) issue_chunk(0, pre);
    for (int c0 = 0; c0 < n_feat; c0 += FC) {
        if (staged) {
            stage_chunk(s_rows, pre);
            __syncthreads();
            if (c0 + FC < n_feat) issue_chunk(c0 + FC, pre);          // lands under this chunk's reduction
        }
        reduce_chunk(c0, s_rows);
        if (staged) __syncthreads();                          // s_rows is overwritten by the next chunk
    }
}


// Fast path for n_feat % FC == 0 (every lane owns a whole 16-byte piece in every chunk): the staged loop has NO branch
// around a vector-memory instruction, so the compiler's s_waitcnt insertion can count -- the distinct source rows of the
// next TWO chunks are in flight in two register sets (loop unrolled by two, both sets statically indexed) while the current
// chunk is reduced from LDS; source row pointers, the row's edge range, its scale and its output pointer are hoisted out of
// the chunk loop.  Same summation order as spmm_tiled_kernel / spmm_csr_kernel: bit-identical results.
template <bool ACCUM>
__global__ void __launch_bounds__(256, 7)   // 72 VGPRs: 7 workgroups per CU (LDS would allow 8)
spmm_tiled_full_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                       const uint16_t* __restrict__ lidx, const float* __restrict__ ew,
                       const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ tile_src,
                       const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
                       int n_rows, int n_feat, int reduce) {
    __shared__ __attribute__((aligned(16))) float s_rows[UMAX * FC];
    struct Edge { float w; int off; };                         // off = local index * FC (float offset of the staged row)
    __shared__ __attribute__((aligned(8))) Edge s_e[EMAX_FULL];
    __shared__ int s_usrc[UMAX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane / LPR, l16 = lane % LPR;               // 8 lanes per 128-byte row piece, 8 row groups per wave
    const int ntiles = (n_rows + TILE_R - 1) / TILE_R;
    const int tile = (int)gte_xcd_remap(blockIdx.x, (unsigned)ntiles);
    const int row0 = tile * TILE_R;
    const int row_end = min(row0 + TILE_R, n_rows);
    const int e0 = indptr[row0], e1 = indptr[row_end];
    const int u0 = tile_ptr[tile], nu = tile_ptr[tile + 1] - u0;
    const int ne = e1 - e0;
    // The tile's sources are addressed as {uniform base = row of the smallest source (tile_src is sorted), 32-bit byte
    // offset}: no 64-bit address arithmetic in the chunk loop.  A tile whose sources span 4 GB or more is gathered directly.
    const int src_lo = nu > 0 ? tile_src[u0] : 0, src_hi = nu > 0 ? tile_src[u0 + nu - 1] : 0;
    const bool staged = (nu <= UMAX) && (ne <= EMAX_FULL) &&
                        ((int64_t)(src_hi - src_lo + 1) * ldx * 4 < ((int64_t)1 << 32));   // block-uniform

    static_assert(TILE_R == 4 * GPW, "one destination row per lane group");
    // Rows of the tile are handed to the lane groups in order of decreasing degree: the groups of a wave run their edge loops
    // in lockstep, so a wave lasts as long as its longest row -- rows of similar degree share a wave (cfg4 backward, the
    // out-edge CSR with degrees 0 ... 24: 872 -> 857 us; tiles of equal degrees, the forward graph, skip the ranking).  Every
    // wave ranks the 32 rows for itself in registers (lane t < 32 holds row t's degree; no LDS, no barrier); a row is still
    // summed by one group in CSR order: same bits as before.
    int r = row0 + wave * GPW + g;
#ifndef GTE_TILED_NO_SORT
    {
        const int t = lane & 31;
        const int d = row0 + t < row_end ? indptr[row0 + t + 1] - indptr[row0 + t] : -1;
        if (!__all(d == __builtin_amdgcn_readfirstlane(d))) {
            int rank = 0;
#pragma unroll
            for (int j = 0; j < TILE_R; ++j) {
                const int dj = __builtin_amdgcn_readlane(d, j);
                rank += (dj > d || (dj == d && j < t)) ? 1 : 0;
            }
            const int by_rank = __builtin_amdgcn_ds_permute(rank << 2, t);        // lane q < 32 <- the row of rank q
            r = row0 + __builtin_amdgcn_ds_bpermute((wave * GPW + g) << 2, by_rank);
        }
    }
#endif
    const bool r_ok = r < row_end;
    int lo = 0, hi = 0;
    if (r_ok) { lo = indptr[r]; hi = indptr[r + 1]; }
    const float scale = (reduce == GTE_REDUCE_MEAN) ? (hi > lo ? 1.0f / (float)(hi - lo) : 0.0f) : 1.0f;
    float* const op = out + (int64_t)(r_ok ? r : row0) * ldo + l16 * 4;

    if (!staged) {                                             // hub tile: gather straight from global memory
        if (r_ok) {
            for (int c0 = 0; c0 < n_feat; c0 += FC) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int e = lo; e < hi; ++e) {
                    const float w0 = ew ? ew[e] : 1.0f;
                    const f4u a = *reinterpret_cast<const f4u*>(x + (int64_t)indices[e] * ldx + c0 + l16 * 4);
                    acc.x = fmaf(w0, a.x, acc.x); acc.y = fmaf(w0, a.y, acc.y);
                    acc.z = fmaf(w0, a.z, acc.z); acc.w = fmaf(w0, a.w, acc.w);
                }
                acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
                if constexpr (ACCUM) {
                    const f4u o = *reinterpret_cast<const f4u*>(op + c0);
                    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                }
                st4_guard(op + c0, acc, 4);
            }
        }
        return;
    }

    for (int i = tid; i < ne; i += 256) { s_e[i].off = (int)lidx[e0 + i] * FC; s_e[i].w = ew ? ew[e0 + i] : 1.0f; }
    for (int i = tid; i < nu; i += 256) s_usrc[i] = tile_src[u0 + i];
    __syncthreads();

    constexpr int NLD = UMAX / (4 * GPW);                      // loads per lane per chunk
    uint32_t so[NLD];                                          // byte offset of this lane's piece of its NLD source rows (slot
#pragma unroll                                                 // clamped: a duplicate load lands in an LDS slot nobody reads)
    for (int k = 0; k < NLD; ++k) {
        const int ur = wave * GPW + g + 4 * GPW * k;
        const int srow = nu > 0 ? s_usrc[min(ur, nu - 1)] : 0;
        so[k] = (uint32_t)(((int64_t)(srow - src_lo) * ldx + l16 * 4) * 4);
    }
    const char* const xb = reinterpret_cast<const char*>(x + (int64_t)src_lo * ldx);      // uniform
    float* const sw = &s_rows[(wave * GPW + g) * FC + l16 * 4];   // slot of load k: sw + k * 4 * GPW * FC
    const float* const sr = &s_rows[l16 * 4];
    const int eb = lo - e0, ee = hi - e0;

    auto issue = [&](float4 (&pre)[NLD], int c0) {
        const char* const xc = xb + (int64_t)c0 * 4;           // uniform
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
#ifdef GTE_TILED_NT_LOAD
            const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(xc + so[k]));
#else
            const f4u t = *reinterpret_cast<const f4u*>(xc + so[k]);
#endif
            pre[k] = make_float4(t.x, t.y, t.z, t.w);
        }
    };
    auto stage = [&](const float4 (&pre)[NLD]) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) *reinterpret_cast<float4*>(sw + k * 4 * GPW * FC) = pre[k];
    };
    auto reduce_chunk = [&](int c0) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = eb;
#ifdef GTE_TILED_NOREDUCE                                      // measurement build: one staged row per destination, no edge loop
        if (e < ee) { const Edge ea = s_e[e]; acc = *reinterpret_cast<const float4*>(sr + ea.off); }
        e = ee;
#endif
        for (; e + 1 < ee; e += 2) {                           // two LDS rows in flight
            const Edge ea = s_e[e], eb2 = s_e[e + 1];
            const float4 a = *reinterpret_cast<const float4*>(sr + ea.off);
            const float4 b = *reinterpret_cast<const float4*>(sr + eb2.off);
            acc.x = fmaf(ea.w, a.x, acc.x); acc.y = fmaf(ea.w, a.y, acc.y);
            acc.z = fmaf(ea.w, a.z, acc.z); acc.w = fmaf(ea.w, a.w, acc.w);
            acc.x = fmaf(eb2.w, b.x, acc.x); acc.y = fmaf(eb2.w, b.y, acc.y);
            acc.z = fmaf(eb2.w, b.z, acc.z); acc.w = fmaf(eb2.w, b.w, acc.w);
        }
        if (e < ee) {
            const Edge ea = s_e[e];
            const float4 a = *reinterpret_cast<const float4*>(sr + ea.off);
            acc.x = fmaf(ea.w, a.x, acc.x); acc.y = fmaf(ea.w, a.y, acc.y);
            acc.z = fmaf(ea.w, a.z, acc.z); acc.w = fmaf(ea.w, a.w, acc.w);
        }
        acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
        if (r_ok) {
            if constexpr (ACCUM) {
                const f4u o = *reinterpret_cast<const f4u*>(op + c0);
                acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            }
            st4_hidden(op + c0, acc);
        }
    };

    // Two register sets, the chunk loop unrolled by two so both are statically indexed.  The only way into the main loop
    // has BOTH sets requested and every issue inside it is unconditional, so the wait in front of a stage is a counted
    // vmcnt(NLD): the other set stays in flight.  Fewer than four chunks, and the last two or three, run a drained tail.
    const int nch = n_feat / FC;
    float4 preA[NLD], preB[NLD];
    int c = 0;
    if (nch >= 4) {
        issue(preA, 0);
        __builtin_amdgcn_sched_barrier(0);                     // set A strictly older than set B (the scheduler would interleave them)
        issue(preB, FC);
        __builtin_amdgcn_sched_barrier(0);
        for (; c + 4 <= nch; c += 2) {
            stage(preA);
            __syncthreads();
            issue(preA, (c + 2) * FC);
            reduce_chunk(c * FC);
            __syncthreads();
            stage(preB);
            __syncthreads();
            issue(preB, (c + 3) * FC);
            reduce_chunk((c + 1) * FC);
            __syncthreads();
        }
    } else {
        issue(preA, 0);
        if (nch > 1) issue(preB, FC);
    }
    stage(preA);
    __syncthreads();
    if (c + 2 < nch) issue(preA, (c + 2) * FC);
    reduce_chunk(c * FC);
    __syncthreads();
    if (c + 1 < nch) {
        stage(preB);
        __syncthreads();
        reduce_chunk((c + 1) * FC);
        __syncthreads();
    }
    if (c + 2 < nch) {
        stage(preA);
        __syncthreads();
        reduce_chunk((c + 2) * FC);
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
    if (n_feat % FC == 0 && n_feat >= 4 * FC) {                // the two-sets-in-flight loop needs four chunks
        if (accumulate)
            hipLaunchKernelGGL(spmm_tiled_full_kernel<true>, grid, block, 0, s, indptr, indices, local_index, eweight,
                               tile_ptr, tile_src, x, ldx, out, ldo, (int)n_rows, (int)n_feat, reduce);
        else
            hipLaunchKernelGGL(spmm_tiled_full_kernel<false>, grid, block, 0, s, indptr, indices, local_index, eweight,
                               tile_ptr, tile_src, x, ldx, out, ldo, (int)n_rows, (int)n_feat, reduce);
        return gte::check_launch("spmm_csr_tiled");
    }
    if (accumulate)
        hipLaunchKernelGGL(spmm_tiled_kernel<true>, grid, block, 0, s, indptr, indices, local_index, eweight, tile_ptr,
                           tile_src, x, ldx, out, ldo, (int)n_rows, (int)n_feat, reduce);
    else
        hipLaunchKernelGGL(spmm_tiled_kernel<false>, grid, block, 0, s, indptr, indices, local_index, eweight, tile_ptr,
                           tile_src, x, ldx, out, ldo, (int)n_rows, (int)n_feat, reduce);
    return gte::check_launch("spmm_csr_tiled");
}
