// Device-side batching of resident page graphs and the edge-weight computation.
//
// replaces, per training step,
//   dgl.batch(train_batch).to(device)                       src/models/model_train.py:297  (block-diagonal union)
//   the host->device copy of every tensor the batch carries  src/models/model_train.py:297-298
// and, at data-preparation time,
//   edata['feat'] = 1 - distance(bbox_u, bbox_v) / max_e distance   src/components/graphs/loader.py:332-344
//   with distance() of src/components/graphs/utils.py:56-88 (integer pixel distances, int(sqrt(.)) on diagonals).
//
// Dataset layout in HBM (all pages of the dataset concatenated once; "resident pages"):
//   node_off[P+1], edge_off[P+1]        int32   page p owns nodes [node_off[p], node_off[p+1]) and CSR entries
//                                               [edge_off[p], edge_off[p+1])
//   indptr_loc[Ntot + P]                int32   per page its own indptr (n_p + 1 entries, starting at 0), packed:
//                                               page p's indptr starts at node_off[p] + p
//   indices_loc[Etot]                   int32   column ids LOCAL to the page
//   weight[Etot]                        f32     per-entry weight in CSR order (may be NULL)
// A batch is a list of page ids; the kernels write the batched CSR (global ids, block diagonal) by pure
// index arithmetic -- no sort: rows of a block-diagonal union are the pages' rows.  HBM-bound integer work,
// one thread per output element, binary search over the (<= a few hundred) batch offsets.
#include "gte_common.h"
#include "batch_assemble.h"

#include <stdlib.h>

namespace {
using gte_asm::seg_of;

__global__ void __launch_bounds__(256)
batch_indptr_kernel(const int32_t* __restrict__ pages, int nb, const int32_t* __restrict__ node_off,
                    const int32_t* __restrict__ b_node_off, const int32_t* __restrict__ b_edge_off,
                    const int32_t* __restrict__ indptr_loc, int32_t* __restrict__ indptr_out, int n_out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r > n_out) return;
    if (r == n_out) { indptr_out[r] = b_edge_off[nb]; return; }
    const int i = seg_of(b_node_off, nb, r);
    const int p = pages[i];
    indptr_out[r] = indptr_loc[node_off[p] + p + (r - b_node_off[i])] + b_edge_off[i];
}

__global__ void __launch_bounds__(256)
batch_edges_kernel(const int32_t* __restrict__ pages, int nb, const int32_t* __restrict__ edge_off,
                   const int32_t* __restrict__ b_node_off, const int32_t* __restrict__ b_edge_off,
                   const int32_t* __restrict__ indices_loc, const float* __restrict__ weight,
                   int32_t* __restrict__ indices_out, float* __restrict__ weight_out, int e_out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= e_out) return;
    const int i = seg_of(b_edge_off, nb, e);
    const int p = pages[i];
    const int64_t s = (int64_t)edge_off[p] + (e - b_edge_off[i]);
    indices_out[e] = indices_loc[s] + b_node_off[i];
    if (weight_out) weight_out[e] = weight ? weight[s] : 1.0f;
}

// rows of a [Ntot, F] matrix (features) or a vector (labels): out[b_node_off[i] + r] = in[node_off[pages[i]] + r]
// one 16-lane group per row chunk; dword granularity (any F, any alignment)
__global__ void __launch_bounds__(256)
batch_rows_kernel(const int32_t* __restrict__ pages, int nb, const int32_t* __restrict__ node_off,
                  const int32_t* __restrict__ b_node_off, const float* __restrict__ in, int64_t ld_in,
                  float* __restrict__ out, int64_t ld_out, int n_out, int f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n_out) return;
    const int i = seg_of(b_node_off, nb, r);
    const float* src = in + (int64_t)(node_off[pages[i]] + (r - b_node_off[i])) * ld_in;
    float* dst = out + (int64_t)r * ld_out;
    for (int j = lane; j < f; j += 64) dst[j] = src[j];
}

// the same with 16-byte accesses (rows need only 4-byte alignment on gfx950): lane l moves chunks l, l + 64, ... of the row,
// every load of a row is requested before its first store; the f % 4 tail goes by dwords.  f <= 1024 NV / 4.
struct __attribute__((packed, aligned(4))) brow4 { float x, y, z, w; };
template <int NV>
__global__ void __launch_bounds__(256)
batch_rows_vec_kernel(const int32_t* __restrict__ pages, int nb, const int32_t* __restrict__ node_off,
                      const int32_t* __restrict__ b_node_off, const float* __restrict__ in, int64_t ld_in,
                      float* __restrict__ out, int64_t ld_out, int n_out, int f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n_out) return;
    const int i = seg_of(b_node_off, nb, r);
    const float* src = in + (int64_t)(node_off[pages[i]] + (r - b_node_off[i])) * ld_in;
    float* dst = out + (int64_t)r * ld_out;
    const int nq = f >> 2;
    brow4 v[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nq) v[u] = *reinterpret_cast<const brow4*>(src + 4 * c);
    }
    const int jt = 4 * nq + lane;
    const float tail = jt < f ? src[jt] : 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nq) *reinterpret_cast<brow4*>(dst + 4 * c) = v[u];
    }
    if (jt < f) dst[jt] = tail;
}

// ---- the whole batch in one launch: the device code lives in batch_assemble.h (the fold launch of a step can carry it too) ----
namespace {
using gte_asm::AssembleArgs;
__global__ void __launch_bounds__(256)
batch_assemble_kernel(const AssembleArgs a, const int feat_wgs, const int rows_per_wg, const int n_out) {
    gte_asm::assemble_block(a, feat_wgs, rows_per_wg, n_out, (int)blockIdx.x);
}
}  // namespace

static thread_local bool g_defer = false;          // gte_batch_assemble_defer

static int batch_assemble_impl(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* b_node_off,
                               const gte_batch_arrays* in_edges, const gte_batch_arrays* out_edges, const float* feat,
                               int64_t ld_feat, int64_t n_cols, float* feat_out, const float* label, float* label_out,
                               int64_t n_out, void* stream, int32_t* row_map, int64_t row_map_pad, int64_t n_res) {
    if (n_batch <= 0 || n_batch > 65535 || n_out < 0 || n_out >= INT32_MAX || (feat && n_cols <= 0))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_assemble: bad sizes");
    if (!pages || !node_off || !b_node_off || ((feat == nullptr) != (feat_out == nullptr)) || (!feat && !row_map))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_assemble: null pointer");
    if (feat && ld_feat != n_cols) return gte::fail(GTE_ERR_UNSUPPORTED, "batch_assemble: resident features must be packed (ld == n_cols)");
    if (row_map && (row_map_pad < 0 || n_res < 0 || n_res > INT32_MAX)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_assemble: row map");
    if ((label == nullptr) != (label_out == nullptr)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_assemble: label / label_out");
    AssembleArgs a = {};
    a.pages = pages; a.node_off = node_off; a.b_node_off = b_node_off; a.nb = (int)n_batch;
    a.feat = feat; a.feat_out = feat_out; a.n_cols = n_cols; a.label = label; a.label_out = label_out;
    a.row_map = row_map; a.row_map_pad = (int)row_map_pad; a.n_res = (int)n_res;
    const gte_batch_arrays* dirs[2] = {in_edges, out_edges};
    for (int d = 0; d < 2; ++d) {
        if (!dirs[d]) continue;
        a.dir[d] = *dirs[d];
        const gte_batch_arrays& g = a.dir[d];
        if (!g.edge_off || !g.indptr_loc || !g.b_edge_off || !g.indptr_out)
            return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_assemble: null CSR pointer");
    }
    if (n_out == 0) return GTE_OK;
    // ~32 KB of feature rows per workgroup (at least one row): ~2 500 workgroups for a 100-page batch of 831-wide rows
    int64_t rows_per_wg = feat ? 32768 / (n_cols * 4) : 1;
    if (rows_per_wg < 1) rows_per_wg = 1;
    static const int forced = GTE_MEASURE_INT("GTE_ASSEMBLE_ROWS", 0);
    if (forced > 0) rows_per_wg = forced;
    const int64_t feat_wgs = feat ? gte::ceil_div(n_out, rows_per_wg) : 0;
    // gte_batch_assemble_defer(1) and an open fold deferral on this stream (the caller is inside a training step, in front of its
    // last GEMM): the job rides the step's fold + optimiser launch as extra workgroups instead of a launch of its own
    if (g_defer) {
        const gte_asm::AssembleJob job = {a, (int)feat_wgs, (int)rows_per_wg, (int)n_out, (int)(feat_wgs + 7 * n_batch)};
        if (gte::defer_assemble(job, gte::as_stream(stream))) return GTE_OK;
    }
    hipLaunchKernelGGL(batch_assemble_kernel, dim3((unsigned)(feat_wgs + 7 * n_batch)), dim3(256), 0, gte::as_stream(stream), a,
                       (int)feat_wgs, (int)rows_per_wg, (int)n_out);
    return gte::check_launch("batch_assemble");
}

extern "C" int gte_batch_assemble_defer(int on) {
    const int was = g_defer ? 1 : 0;
    g_defer = on != 0;
    return was;
}

extern "C" int gte_batch_assemble(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* b_node_off,
                                  const gte_batch_arrays* in_edges, const gte_batch_arrays* out_edges, const float* feat,
                                  int64_t ld_feat, int64_t n_cols, float* feat_out, const float* label, float* label_out,
                                  int64_t n_out, void* stream) {
    if (!feat || !feat_out) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_assemble: null pointer");
    return batch_assemble_impl(pages, n_batch, node_off, b_node_off, in_edges, out_edges, feat, ld_feat, n_cols, feat_out, label, label_out,
                               n_out, stream, nullptr, 0, 0);
}

// ... with a ROW MAP instead of (feat == NULL) or next to the copied feature rows: row_map[r] = resident row of batch row r for
// r < n_out, and row_map[n_out .. n_out + row_map_pad) = n_res_rows (a row past the resident matrix).  The planes GEMMs of the
// input layer read the resident feature image through the map (gte_gemm_p3_nt_rows / gte_gemm_p3_tn_rows): a batch then costs
// no feature traffic at all.
extern "C" int gte_batch_assemble_rows(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* b_node_off,
                                       const gte_batch_arrays* in_edges, const gte_batch_arrays* out_edges, const float* feat,
                                       int64_t ld_feat, int64_t n_cols, float* feat_out, const float* label, float* label_out,
                                       int64_t n_out, int32_t* row_map, int64_t row_map_pad, int64_t n_res_rows, void* stream) {
    if (!row_map) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_assemble_rows: null row map");
    return batch_assemble_impl(pages, n_batch, node_off, b_node_off, in_edges, out_edges, feat, ld_feat, n_cols, feat_out, label, label_out,
                               n_out, stream, row_map, row_map_pad, n_res_rows);
}

// ---- edge weights from word boxes -----------------------------------------------------------------
__device__ __forceinline__ int box_distance(const int4 a, const int4 b) {
    // graphs/utils.py:56-88: 0 if the boxes intersect (touching counts), the axis gap if they face each other,
    // else int(sqrt(dx^2 + dy^2)) between the nearest corners.  Boxes are (x0, y0, x1, y1) integers.
    const int dx = max(max(b.x - a.z, a.x - b.z), 0);
    const int dy = max(max(b.y - a.w, a.y - b.w), 0);
    if (dx > 0 && dy > 0) return (int)sqrt((double)dx * dx + (double)dy * dy);
    return max(dx, dy);
}

__global__ void __launch_bounds__(256)
edge_dist_kernel(const int32_t* __restrict__ bbox, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                 const int32_t* __restrict__ graph_of_node, int32_t* __restrict__ dist, int32_t* __restrict__ gmax,
                 int64_t n_edges) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    const int u = src[e], v = dst[e];
    const int4 a = *reinterpret_cast<const int4*>(bbox + (int64_t)u * 4);
    const int4 b = *reinterpret_cast<const int4*>(bbox + (int64_t)v * 4);
    const int d = box_distance(a, b);
    dist[e] = d;
    atomicMax(&gmax[graph_of_node[v]], d);        // integer max: order-independent, deterministic
}

__global__ void __launch_bounds__(256)
edge_weight_kernel(const int32_t* __restrict__ dist, const int32_t* __restrict__ dst,
                   const int32_t* __restrict__ graph_of_node, const int32_t* __restrict__ gmax,
                   float* __restrict__ w, int64_t n_edges) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    const int m = gmax[graph_of_node[dst[e]]];
    // the reference divides by max(distances) even when it is 0 (loader.py:341-342: ZeroDivisionError); a page
    // whose edges all have distance 0 gets weight 1 here -- documented deviation
    w[e] = m > 0 ? (float)(1.0 - (double)dist[e] / (double)m) : 1.0f;
}

}  // namespace

extern "C" int gte_batch_csr(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* edge_off,
                             const int32_t* b_node_off, const int32_t* b_edge_off, const int32_t* indptr_loc,
                             const int32_t* indices_loc, const float* weight, int32_t* indptr_out,
                             int32_t* indices_out, float* weight_out, int64_t n_out, int64_t e_out, void* stream) {
    if (n_batch <= 0 || n_out < 0 || e_out < 0 || n_out >= INT32_MAX || e_out >= INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_csr: bad sizes");
    if (!pages || !node_off || !edge_off || !b_node_off || !b_edge_off || !indptr_loc || !indptr_out)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_csr: null pointer");
    if (e_out > 0 && (!indices_loc || !indices_out)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_csr: null indices");
    hipStream_t s = gte::as_stream(stream);
    hipLaunchKernelGGL(batch_indptr_kernel, dim3((unsigned)gte::ceil_div(n_out + 1, 256)), dim3(256), 0, s, pages,
                       (int)n_batch, node_off, b_node_off, b_edge_off, indptr_loc, indptr_out, (int)n_out);
    if (e_out > 0)
        hipLaunchKernelGGL(batch_edges_kernel, dim3((unsigned)gte::ceil_div(e_out, 256)), dim3(256), 0, s, pages,
                           (int)n_batch, edge_off, b_node_off, b_edge_off, indices_loc, weight, indices_out, weight_out,
                           (int)e_out);
    return gte::check_launch("batch_csr");
}

extern "C" int gte_batch_rows(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* b_node_off,
                              const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t n_out,
                              int64_t n_cols, void* stream) {
    if (n_batch <= 0 || n_out < 0 || n_cols <= 0 || n_out >= INT32_MAX || n_cols > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_rows: bad sizes");
    if (n_out == 0) return GTE_OK;
    if (!pages || !node_off || !b_node_off || !in || !out) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_rows: null pointer");
    if (ld_in < n_cols || ld_out < n_cols) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "batch_rows: ld < n_cols");
    const dim3 grid((unsigned)gte::ceil_div(n_out, 4)), block(256);
    hipStream_t s = gte::as_stream(stream);
#define GTE_BROWS(NV)                                                                                                  \
    hipLaunchKernelGGL((batch_rows_vec_kernel<NV>), grid, block, 0, s, pages, (int)n_batch, node_off, b_node_off, in, ld_in, \
                       out, ld_out, (int)n_out, (int)n_cols)
    if (n_cols >= 16 && n_cols <= 256) GTE_BROWS(1);
    else if (n_cols > 256 && n_cols <= 512) GTE_BROWS(2);
    else if (n_cols > 512 && n_cols <= 1024) GTE_BROWS(4);
    else
        hipLaunchKernelGGL(batch_rows_kernel, grid, block, 0, s, pages, (int)n_batch, node_off, b_node_off, in, ld_in, out,
                           ld_out, (int)n_out, (int)n_cols);
#undef GTE_BROWS
    return gte::check_launch("batch_rows");
}

extern "C" int64_t gte_edge_weights_workspace_bytes(int64_t n_edges, int64_t n_graphs) {
    return gte::round_up((n_edges > 0 ? n_edges : 1) * 4, 256) + gte::round_up((n_graphs > 0 ? n_graphs : 1) * 4, 256);
}

extern "C" int gte_edge_weights_bbox(const int32_t* bbox, const int32_t* src, const int32_t* dst,
                                     const int32_t* graph_of_node, int64_t n_edges, int64_t n_graphs, float* weight,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_edges < 0 || n_graphs <= 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "edge_weights_bbox: bad sizes");
    if (n_edges == 0) return GTE_OK;
    if (!bbox || !src || !dst || !graph_of_node || !weight || !workspace)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "edge_weights_bbox: null pointer");
    if (((uintptr_t)bbox & 15) != 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "edge_weights_bbox: bbox must be 16-byte aligned");
    if (workspace_bytes < gte_edge_weights_workspace_bytes(n_edges, n_graphs))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "edge_weights_bbox: workspace too small");
    hipStream_t s = gte::as_stream(stream);
    int32_t* dist = reinterpret_cast<int32_t*>(workspace);
    int32_t* gmax = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + gte::round_up(n_edges * 4, 256));
    if (hipMemsetAsync(gmax, 0, (size_t)n_graphs * 4, s) != hipSuccess)
        return gte::fail(GTE_ERR_LAUNCH, "edge_weights_bbox: memset failed");
    const unsigned blocks = (unsigned)gte::ceil_div(n_edges, 256);
    hipLaunchKernelGGL(edge_dist_kernel, dim3(blocks), dim3(256), 0, s, bbox, src, dst, graph_of_node, dist, gmax, n_edges);
    hipLaunchKernelGGL(edge_weight_kernel, dim3(blocks), dim3(256), 0, s, dist, dst, graph_of_node, gmax, weight, n_edges);
    return gte::check_launch("edge_weights_bbox");
}

// ---- BBOX node features on the device ---------------------------------------------
// replaces src/components/nlp/bbox.py:49-124 (Bbox.__call__), called per batch through _generate_features at
// src/models/model_train.py:293: per word 9 geometry values
//   [w, h, cx, cy, w*h, x0, y0, x1, y1],  cx = x1 - int(w/2), cy = y1 - int(h/2)        (bbox.py:49-54)
// and the 4-bin content histogram [letters, digits, others, empty] of the word's characters (bbox.py:56-107):
// fractions in float64, the largest bin absorbs 1 - sum so the bins add up to exactly 1, empty text -> [0,0,0,1].
// The character classes are counted on the host (str.isalpha / str.isdigit are Unicode tables); everything
// arithmetic is here.  Output = float32 of the float64 values, as `.float()` does at model_train.py:296.
namespace {
__global__ void __launch_bounds__(256)
bbox_features_kernel(const int32_t* __restrict__ bbox, const int32_t* __restrict__ counts, float* __restrict__ out,
                     int64_t ldo, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x0 = bbox[i * 4 + 0], y0 = bbox[i * 4 + 1], x1 = bbox[i * 4 + 2], y1 = bbox[i * 4 + 3];
    const int w = x1 - x0, h = y1 - y0;
    // Python int(w/2): true division then truncation toward zero
    const int cx = x1 - (int)((double)w / 2.0), cy = y1 - (int)((double)h / 2.0);
    float* o = out + i * ldo;
    o[0] = (float)w; o[1] = (float)h; o[2] = (float)cx; o[3] = (float)cy;
    o[4] = (float)((double)w * (double)h);
    o[5] = (float)x0; o[6] = (float)y0; o[7] = (float)x1; o[8] = (float)y1;
    const int na = counts[i * 3 + 0], nd = counts[i * 3 + 1], no = counts[i * 3 + 2];
    const int ns = na + nd + no;
    double hst[4] = {0.0, 0.0, 0.0, 0.0};
    if (ns != 0) {
        hst[0] = (double)na / (double)ns;
        hst[1] = (double)nd / (double)ns;
        hst[2] = (double)no / (double)ns;
        const double sum = ((hst[0] + hst[1]) + hst[2]) + hst[3];       // Python sum(): left to right from 0
        if (sum != 1.0) {
            const double diff = 1.0 - sum;
            int im = 0;                                                   // list.index(max(...)): first maximum
            for (int j = 1; j < 4; ++j) if (hst[j] > hst[im]) im = j;
            hst[im] = hst[im] + diff;
        }
    }
    if (hst[0] == 0.0 && hst[1] == 0.0 && hst[2] == 0.0) hst[3] = 1.0;
    for (int j = 0; j < 4; ++j) o[9 + j] = (float)hst[j];
}
}  // namespace

extern "C" int gte_bbox_features(const int32_t* bbox, const int32_t* char_counts, float* out, int64_t ldo,
                                 int64_t n_nodes, void* stream) {
    if (n_nodes < 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "bbox_features: n < 0");
    if (n_nodes == 0) return GTE_OK;
    if (!bbox || !char_counts || !out) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "bbox_features: null pointer");
    if (ldo < 13) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "bbox_features: ldo < 13");
    hipLaunchKernelGGL(bbox_features_kernel, dim3((unsigned)gte::ceil_div(n_nodes, 256)), dim3(256), 0,
                       gte::as_stream(stream), bbox, char_counts, out, ldo, n_nodes);
    return gte::check_launch("bbox_features");
}
