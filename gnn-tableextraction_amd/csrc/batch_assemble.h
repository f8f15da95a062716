// Batch assembly as DEVICE code shared by two launches: gte_batch_assemble's own kernel (csrc/batch_ops.hip) and the fold +
// optimiser launch at the end of a training step (csrc/gte_core.hip), which can carry the NEXT batch's assembly as extra
// workgroups (gte_batch_assemble_defer: a row-map batch is 7 us of page table, CSRs and labels -- latency, not bytes -- which
// then cost nothing on the step's stream).
#pragma once
#include "gte_common.h"

namespace gte_asm {

// largest i in [0, nb) with off[i] <= x   (off is ascending, off[0] = 0)
__device__ __forceinline__ int seg_of(const int32_t* __restrict__ off, int nb, int x) {
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- the whole batch in one launch -----------------------------------------------------------------------------
// One run = n consecutive dwords src -> dst (+ an integer constant for index arrays).  The workgroups of a page split
// every run in 16-byte pieces aligned on the DESTINATION (head / tail dwords by single lanes); four pieces per lane are
// requested before the first store.
struct __attribute__((packed, aligned(4))) u4 { uint32_t x, y, z, w; };

template <bool ADD>
__device__ __forceinline__ void copy_run(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n, uint32_t add,
                                         int worker, int workers) {
    if (n <= 0) return;
    int64_t head = (int64_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15) >> 2;
    if (head > n) head = n;
    const int64_t body = (n - head) >> 2, tail0 = head + 4 * body;
    const int t = worker * 256 + (int)threadIdx.x, stride = workers * 256;
    if (worker == 0) {
        if ((int64_t)threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x] + (ADD ? add : 0u);
        const int64_t j = tail0 + (threadIdx.x - 64);
        if (threadIdx.x >= 64 && j < n) dst[j] = src[j] + (ADD ? add : 0u);
    }
    const u4* s4 = reinterpret_cast<const u4*>(src + head);        // 4-byte aligned source: gfx950 unaligned access mode
    uint4* d4 = reinterpret_cast<uint4*>(dst + head);
    int64_t c = t;
    for (; c + 3 * (int64_t)stride < body; c += 4 * (int64_t)stride) {
        u4 v0 = s4[c], v1 = s4[c + stride], v2 = s4[c + 2 * (int64_t)stride], v3 = s4[c + 3 * (int64_t)stride];
        if (ADD) {
            v0.x += add; v0.y += add; v0.z += add; v0.w += add; v1.x += add; v1.y += add; v1.z += add; v1.w += add;
            v2.x += add; v2.y += add; v2.z += add; v2.w += add; v3.x += add; v3.y += add; v3.z += add; v3.w += add;
        }
        d4[c] = make_uint4(v0.x, v0.y, v0.z, v0.w);
        d4[c + stride] = make_uint4(v1.x, v1.y, v1.z, v1.w);
        d4[c + 2 * (int64_t)stride] = make_uint4(v2.x, v2.y, v2.z, v2.w);
        d4[c + 3 * (int64_t)stride] = make_uint4(v3.x, v3.y, v3.z, v3.w);
    }
    for (; c < body; c += stride) {
        u4 v = s4[c];
        if (ADD) { v.x += add; v.y += add; v.z += add; v.w += add; }
        d4[c] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

struct AssembleArgs {
    const int32_t* pages; const int32_t* node_off; const int32_t* b_node_off;
    gte_batch_arrays dir[2];
    const float* feat; float* feat_out; int64_t n_cols;
    const float* label; float* label_out;
    int nb;
    int32_t* row_map; int row_map_pad; int n_res;      // optional: resident row of every batch row (+ row_map_pad entries = n_res)
};


// a batch's assembly as data: what a launch needs (blocks == 0: none)
struct AssembleJob { AssembleArgs a; int feat_wgs, rows_per_wg, n_out, blocks; };

// 1-D grid: [0, feat_wgs): workgroup b moves the feature rows [b, b + 1) * rows_per_wg of the BATCH -- equal bytes per
// workgroup whatever the page sizes (pages vary 20 ... 2000 words; a fixed number of workgroups per page left the
// largest page's workgroups running 10x longer than the rest: 45 us for 162 MB); the pages a range overlaps are found with one
// binary search per workgroup.  [feat_wgs, feat_wgs + 7 * pages): one small run of one page each (label; per CSR direction
// indptr / indices / weights) -- one dependent chain (page id -> offsets -> loads -> stores) per workgroup.
__device__ __forceinline__ void assemble_block(const AssembleArgs& a, const int feat_wgs, const int rows_per_wg, const int n_out,
                                               const int block) {
    if (block < feat_wgs) {
        const int R0 = block * rows_per_wg, R1 = min(n_out, R0 + rows_per_wg);
        for (int i = seg_of(a.b_node_off, a.nb, R0); i < a.nb; ++i) {
            const int pb = a.b_node_off[i], pe = a.b_node_off[i + 1];
            if (pb >= R1) break;
            const int lo = max(R0, pb), hi = min(R1, pe);
            if (hi <= lo) continue;                                  // empty page
            const int64_t src_row = (int64_t)a.node_off[a.pages[i]] + (lo - pb);
            copy_run<false>(reinterpret_cast<const uint32_t*>(a.feat + src_row * a.n_cols),
                            reinterpret_cast<uint32_t*>(a.feat_out + (int64_t)lo * a.n_cols), (int64_t)(hi - lo) * a.n_cols, 0u, 0, 1);
        }
        return;
    }
    const int u = block - feat_wgs;
    const int i = u / 7, job = u % 7;                        // job 0: label (+ row map); 1..3: in-edge CSR; 4..6: out-edge CSR
    const int p = a.pages[i];
    const int64_t r0 = a.node_off[p], n_i = a.node_off[p + 1] - r0;
    const int64_t o0 = a.b_node_off[i];
    if (job == 0) {
        if (a.label)
            copy_run<false>(reinterpret_cast<const uint32_t*>(a.label + r0), reinterpret_cast<uint32_t*>(a.label_out + o0), n_i, 0u, 0, 1);
        if (a.row_map) {
            // batch row o0 + r  <-  resident row r0 + r: the planes GEMMs read the resident feature image through this map
            // instead of a per-batch copy of the rows; the entries past the batch name a row past the resident image
            // (out of the GEMM's buffer window: zeros)
            for (int64_t r = threadIdx.x; r < n_i; r += 256) a.row_map[o0 + r] = (int32_t)(r0 + r);
            if (i == a.nb - 1)
                for (int r = threadIdx.x; r < a.row_map_pad; r += 256) a.row_map[n_out + r] = a.n_res;
        }
        return;
    }
    const gte_batch_arrays& g = a.dir[(job - 1) / 3];
    if (!g.indptr_out) return;
    const int64_t e0 = g.edge_off[p], eo = g.b_edge_off[i], e_i = g.b_edge_off[i + 1] - eo;
    const int what = (job - 1) % 3;
    if (what == 0)
        copy_run<true>(reinterpret_cast<const uint32_t*>(g.indptr_loc + r0 + p), reinterpret_cast<uint32_t*>(g.indptr_out + o0),
                       n_i + (i == a.nb - 1 ? 1 : 0), (uint32_t)eo, 0, 1);
    else if (what == 1)
        copy_run<true>(reinterpret_cast<const uint32_t*>(g.indices_loc + e0), reinterpret_cast<uint32_t*>(g.indices_out + eo), e_i,
                       (uint32_t)o0, 0, 1);
    else if (g.weight_out) {
        if (g.weight)
            copy_run<false>(reinterpret_cast<const uint32_t*>(g.weight + e0), reinterpret_cast<uint32_t*>(g.weight_out + eo), e_i, 0u,
                            0, 1);
        else
            for (int64_t j = threadIdx.x; j < e_i; j += 256) g.weight_out[eo + j] = 1.0f;
    }
}

}  // namespace gte_asm

namespace gte {
// Inside an open fold deferral on `stream` (and with gte_batch_assemble_defer(1) in force): keep the job for the deferral's flush
// launch instead of launching it now.  false: not taken (no deferral, another stream, a job already waiting) -- launch it yourself.
bool defer_assemble(const gte_asm::AssembleJob& job, hipStream_t stream);
}  // namespace gte
