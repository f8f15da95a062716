// k-NN page-graph construction on the device: word boxes -> the in-edge CSR the model aggregates over.
//
// replaces, per PDF page (reference: Python loops over per-pixel projection lists, O(n^2)-ish per page on the host),
//   src/components/graphs/builder.py:383-395   projections: node j owns pixel columns [x0, x1) and rows [y0, y1)
//   src/components/graphs/builder.py:240-292   knn(): grow a window around the node (multiplier m = 2, 3, ... 99; a box wider
//                                              than tall grows by w m / 4 horizontally and h m vertically, else w m and h m / 4)
//                                              until it overlaps >= k boxes (the node itself counts); of the LAST window's boxes
//                                              take the k nearest by distance() (graphs/utils.py:56-88); keep those <= max_dist;
//                                              edge neighbour -> node; skip an edge whose reverse an EARLIER node added
//   src/components/graphs/loader.py:313-320    dgl.to_simple + dgl.to_bidirected
//   src/components/graphs/builder.py:567-582   fast_remove_islands: TEXT nodes that reach no non-TEXT node by a walk of exactly
//                                              `range_island` steps (dgl.khop_adj = A^k) on the bidirected graph
//
// Equal-distance candidates: the reference orders them by CPython set iteration + numpy's unstable argsort; here (and in
// oracle/knn_graph.py, which is pinned on the reference's own output) ties break by (distance, node id).
//
// Structure.  The window test is monotone in m, so "the smallest m at which box j enters node i's window" (m_ij, by binary
// search over m: 7 probes of integer arithmetic) replaces the reference's re-scan per multiplier: the window stops at m* = the
// k-th smallest m_ij, its candidates are {j : m_ij <= m*}.  One thread per node, the page's boxes staged in LDS (a page has at
// most a few thousand words), two sweeps over the page (m*, then the k nearest candidates kept sorted by (distance, id) in
// registers).  Pure integer work, no atomics, no sort: deterministic.  The symmetric closure is enumerated per destination in
// ascending source id (a in-neighbour of b  <=>  a in sel(b) or b in sel(a)), which IS the sorted, duplicate-free in-edge CSR.
#include "gte_common.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int KNN_KMAX = 16;          // k of PREPROCESS.k (default 5)
constexpr int KNN_PAGE_MAX = 4096;    // boxes of one page staged in LDS (64 KB)
constexpr int M_NONE = 100;           // "never enters the window" (the reference stops at multiplier 99)

struct Box { int x0, y0, x1, y1; };

__device__ __forceinline__ int box_dist(const Box a, const Box b) {          // graphs/utils.py:56-88 (see batch_ops.hip)
    const int dx = max(max(b.x0 - a.x1, a.x0 - b.x1), 0);
    const int dy = max(max(b.y0 - a.y1, a.y0 - b.y1), 0);
    if (dx > 0 && dy > 0) return (int)sqrt((double)dx * dx + (double)dy * dy);
    return max(dx, dy);
}

// Does a box with pixels [lo, hi) own a projection slot inside [w0, w1)?  builder.py:386-394 files pixel hp under slot hp, pixels
// >= extent under the LAST slot (`if hp >= width: hp = width - 1`) and -- Python indexing -- a negative pixel hp under slot
// extent + hp.  For boxes wholly on the canvas this is the plain interval overlap.
__device__ __forceinline__ bool projection_hits(int lo, int hi, int w0, int w1, int extent) {
    const bool inside = max(max(lo, 0), w0) < min(min(hi, extent), w1);
    const bool over = hi > max(lo, extent) && w0 <= extent - 1 && extent - 1 < w1;
    const bool neg = lo < 0 && max(lo + extent, w0) < min(min(hi, 0) + extent, w1);
    return inside || over || neg;
}

// does box b have a pixel column in [wx0, wx1) and a pixel row in [wy0, wy1) of node a's window of multiplier m?
__device__ __forceinline__ bool in_window(const Box a, const Box b, int m, int width, int height) {
    const int w = a.x1 - a.x0, h = a.y1 - a.y0;
    // Python: int(w * m / 4) with non-negative operands == (w * m) >> 2
    const int ho = w > h ? (w * m) >> 2 : w * m;
    const int vo = w > h ? h * m : (h * m) >> 2;
    const int wx0 = max(a.x0 - ho, 0), wy0 = max(a.y0 - vo, 0);
    const int wx1 = min(max(a.x1 + ho, 0), width), wy1 = min(max(a.y1 + vo, 0), height);
    return projection_hits(b.x0, b.x1, wx0, wx1, width) && projection_hits(b.y0, b.y1, wy0, wy1, height);
}

__device__ __forceinline__ int enter_multiplier(const Box a, const Box b, int width, int height) {
    if (!in_window(a, b, 99, width, height)) return M_NONE;
    int lo = 2, hi = 99;                               // smallest m in [2, 99] with in_window (monotone in m)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (in_window(a, b, mid, width, height)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// grid = (ceil(max page size / 256), pages).  sel[i * k + s] = global id of the s-th nearest selected neighbour of node i, -1 padded.
__global__ void __launch_bounds__(256)
knn_select_kernel(const int32_t* __restrict__ bbox, const int32_t* __restrict__ node_off, const int32_t* __restrict__ page_size,
                  int k, int max_dist, int32_t* __restrict__ sel) {
    extern __shared__ Box boxes[];
    const int page = blockIdx.y;
    const int n0 = node_off[page], np = node_off[page + 1] - n0;
    if ((int)blockIdx.x * 256 >= np) return;
    const int width = page_size[2 * page], height = page_size[2 * page + 1];
    for (int j = threadIdx.x; j < np; j += 256) {
        const int4 v = *reinterpret_cast<const int4*>(bbox + (int64_t)(n0 + j) * 4);
        boxes[j] = Box{v.x, v.y, v.z, v.w};
    }
    __syncthreads();
    const int li = blockIdx.x * 256 + threadIdx.x;
    if (li >= np) return;
    const Box a = boxes[li];
    // sweep 1: the k smallest enter-multipliers (the node itself counts when it has a non-empty projection)
    int ms[KNN_KMAX];
#pragma unroll
    for (int s = 0; s < KNN_KMAX; ++s) ms[s] = M_NONE;
    for (int j = 0; j < np; ++j) {
        int m = enter_multiplier(a, boxes[j], width, height);
#pragma unroll
        for (int s = 0; s < KNN_KMAX; ++s) {              // insertion into the ascending list (registers, static indices)
            const int lo = min(ms[s], m);
            m = max(ms[s], m);
            ms[s] = lo;
        }
    }
    int mstar = M_NONE;
#pragma unroll
    for (int s = 0; s < KNN_KMAX; ++s) if (s == k - 1) mstar = ms[s];
    if (mstar >= M_NONE) mstar = 99;                      // never k boxes: the loop ran to its last multiplier
    // sweep 2: of the boxes inside window(m*), the k nearest by (distance, id), the node itself excluded
    int bd[KNN_KMAX], bi[KNN_KMAX];
#pragma unroll
    for (int s = 0; s < KNN_KMAX; ++s) { bd[s] = INT32_MAX; bi[s] = INT32_MAX; }
    for (int j = 0; j < np; ++j) {
        if (j == li || !in_window(a, boxes[j], mstar, width, height)) continue;
        int d = box_dist(a, boxes[j]), id = j;
#pragma unroll
        for (int s = 0; s < KNN_KMAX; ++s) {
            const bool less = d < bd[s] || (d == bd[s] && id < bi[s]);
            const int td = less ? bd[s] : d, ti = less ? bi[s] : id;
            bd[s] = less ? d : bd[s];
            bi[s] = less ? id : bi[s];
            d = td; id = ti;
        }
    }
    int32_t* out = sel + (int64_t)(n0 + li) * k;
#pragma unroll
    for (int s = 0; s < KNN_KMAX; ++s)
        if (s < k) out[s] = (bi[s] != INT32_MAX && bd[s] <= max_dist) ? n0 + bi[s] : -1;
}

__device__ __forceinline__ bool selects(const int32_t* __restrict__ sel, int k, int node, int target) {
    bool hit = false;
    for (int s = 0; s < k; ++s) hit |= sel[(int64_t)node * k + s] == target;
    return hit;
}

// is a -> b an edge of the graph?  bidirectional: symmetric closure.  Directed: builder.py:289 -- the edge a -> b comes from b's
// list and is skipped iff its reverse b -> a was added while an earlier node was processed: a < b and b in sel(a).
__device__ __forceinline__ bool has_edge(const int32_t* __restrict__ sel, int k, int a, int b, int bidirectional) {
    const bool ab = selects(sel, k, b, a);                 // a in sel(b)
    if (bidirectional) return ab || selects(sel, k, a, b);
    return ab && !(a < b && selects(sel, k, a, b));
}

// mode 0: deg[b] = number of in-neighbours; mode 1: indices[indptr[b] ..] = the in-neighbours in ascending id, dst_of[..] = b
__global__ void __launch_bounds__(256)
knn_edges_kernel(const int32_t* __restrict__ sel, const int32_t* __restrict__ node_off, const int32_t* __restrict__ page_of_node,
                 int k, int bidirectional, int mode, int32_t* __restrict__ deg, const int32_t* __restrict__ indptr,
                 int32_t* __restrict__ indices, int32_t* __restrict__ dst_of, int n_nodes) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_nodes) return;
    const int page = page_of_node[b];
    const int n0 = node_off[page], n1 = node_off[page + 1];
    int c = 0;
    const int base = mode ? indptr[b] : 0;
    for (int a = n0; a < n1; ++a) {
        if (has_edge(sel, k, a, b, bidirectional)) {
            if (mode) { indices[base + c] = a; dst_of[base + c] = b; }
            ++c;
        }
    }
    if (!mode) deg[b] = c;
}

// one round of "some neighbour reaches": next[v] = OR over in-neighbours u of cur[u]  (graph symmetric)
__global__ void __launch_bounds__(256)
reach_round_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, const uint8_t* __restrict__ cur,
                   uint8_t* __restrict__ next, int n) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    uint8_t r = 0;
    for (int e = indptr[v]; e < indptr[v + 1]; ++e) r |= cur[indices[e]];
    next[v] = r;
}

__global__ void __launch_bounds__(256)
island_init_kernel(const int32_t* __restrict__ label, int text, uint8_t* __restrict__ cur, int n) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < n) cur[v] = label[v] != text;
}

__global__ void __launch_bounds__(256)
island_final_kernel(const int32_t* __restrict__ label, int text, const uint8_t* __restrict__ reach, uint8_t* __restrict__ island, int n) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < n) island[v] = (!reach[v]) && label[v] == text;
}

// ---- visibility graph (builder.py:294-379): nearest visible box per direction, crossing vertical edges removed ---------------
// sel[i * 4 + {0, 1, 2, 3}] = global id of node i's top / right / bottom / left neighbour, -1 none.  One thread per node scans
// the page's boxes IN INDEX ORDER (the reference's updates are order dependent: a box intersecting the node's box takes the
// top / bottom slot at distance 0 unconditionally, later candidates must beat the current distance strictly; the top and
// right slots also require height / 2 (width / 2) > current distance, the bottom and left slots do not).  Comparisons of
// box centres and of the halves run on doubled integers (exact).  oracle/visibility_graph.py is pinned on the reference's
// own output.
__global__ void __launch_bounds__(256)
vis_select_kernel(const int32_t* __restrict__ bbox, const int32_t* __restrict__ node_off, const int32_t* __restrict__ page_size,
                  int max_dist, int32_t* __restrict__ sel) {
    extern __shared__ Box boxes[];
    const int page = blockIdx.y;
    const int n0 = node_off[page], np = node_off[page + 1] - n0;
    if ((int)blockIdx.x * 256 >= np) return;
    const int width = page_size[2 * page], height = page_size[2 * page + 1];
    for (int j = threadIdx.x; j < np; j += 256) {
        const int4 v = *reinterpret_cast<const int4*>(bbox + (int64_t)(n0 + j) * 4);
        boxes[j] = Box{v.x, v.y, v.z, v.w};
    }
    __syncthreads();
    const int li = blockIdx.x * 256 + threadIdx.x;
    if (li >= np) return;
    const Box a = boxes[li];
    const int acx = a.x0 + a.x1, acy = a.y0 + a.y1;
    int nb0 = -1, nb1 = -1, nb2 = -1, nb3 = -1;
    int c0 = max_dist, c1 = max_dist, c2 = max_dist, c3 = max_dist;
    for (int j = 0; j < np; ++j) {
        if (j == li) continue;
        const Box b = boxes[j];
        const int bcx = b.x0 + b.x1, bcy = b.y0 + b.y1;
        const bool top = bcy < acy, bottom = acy < bcy, right = acx < bcx, left = bcx < acx;
        const bool vp = a.x0 <= b.x1 && b.x0 <= a.x1, hp = a.y0 <= b.y1 && b.y0 <= a.y1;
        if (vp && hp) {
            if (top) { nb0 = j; c0 = 0; }
            else if (bottom) { nb2 = j; c2 = 0; }
        } else if (vp) {
            const int dt = a.y0 - b.y1, db = b.y0 - a.y1;
            if (top && height > 2 * c0 && c0 > dt) { nb0 = j; c0 = dt; }
            else if (bottom && c2 > db) { nb2 = j; c2 = db; }
        } else if (hp) {
            const int dr = b.x0 - a.x1, dl = a.x0 - b.x1;
            if (right && width > 2 * c1 && c1 > dr) { nb1 = j; c1 = dr; }
            else if (left && c3 > dl) { nb3 = j; c3 = dl; }
        }
    }
    int4 o;
    o.x = nb0 < 0 ? -1 : n0 + nb0; o.y = nb1 < 0 ? -1 : n0 + nb1; o.z = nb2 < 0 ? -1 : n0 + nb2; o.w = nb3 < 0 ? -1 : n0 + nb3;
    *reinterpret_cast<int4*>(sel + (int64_t)(n0 + li) * 4) = o;
}

struct VisNode { int cx, cy, right, left; };              // doubled centre, horizontal neighbours (page-local ids, -1 none)

__device__ __forceinline__ bool vis_ccw(int ax, int ay, int bx, int by, int cx, int cy) {
    return (long long)(cy - ay) * (bx - ax) > (long long)(by - ay) * (cx - ax);
}
// builder.py:358-363: do the segments AB and CD cross?
__device__ __forceinline__ bool vis_cross(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy) {
    return vis_ccw(ax, ay, cx, cy, dx, dy) != vis_ccw(bx, by, cx, cy, dx, dy) &&
           vis_ccw(ax, ay, bx, by, cx, cy) != vis_ccw(ax, ay, bx, by, dx, dy);
}

// remove_vertical(): node i's top edge (top -> i) and bottom edge (i -> bottom) are dropped when their centre-to-centre segment
// crosses the segment of any horizontal edge of the page whose END point differs from the vertical edge's START point.
__global__ void __launch_bounds__(256)
vis_prune_kernel(const int32_t* __restrict__ bbox, const int32_t* __restrict__ node_off, int32_t* __restrict__ sel) {
    extern __shared__ VisNode vn[];
    const int page = blockIdx.y;
    const int n0 = node_off[page], np = node_off[page + 1] - n0;
    if ((int)blockIdx.x * 256 >= np) return;
    for (int j = threadIdx.x; j < np; j += 256) {
        const int4 v = *reinterpret_cast<const int4*>(bbox + (int64_t)(n0 + j) * 4);
        const int4 sj = *reinterpret_cast<const int4*>(sel + (int64_t)(n0 + j) * 4);
        vn[j] = VisNode{v.x + v.z, v.y + v.w, sj.y < 0 ? -1 : sj.y - n0, sj.w < 0 ? -1 : sj.w - n0};
    }
    __syncthreads();
    const int li = blockIdx.x * 256 + threadIdx.x;
    if (li >= np) return;
    int32_t* mine = sel + (int64_t)(n0 + li) * 4;
    const int top = mine[0], bot = mine[2];
    for (int side = 0; side < 2; ++side) {
        const int other = side == 0 ? top : bot;
        if (other < 0) continue;
        const int s = side == 0 ? other - n0 : li, d = side == 0 ? li : other - n0;       // top -> i, i -> bottom
        const int v1x = vn[s].cx, v1y = vn[s].cy, v2x = vn[d].cx, v2y = vn[d].cy;
        bool crossed = false;
        for (int j = 0; j < np && !crossed; ++j) {
            const VisNode h = vn[j];
            if (h.left >= 0) {                               // left -> j: h1 = centre(left), h2 = centre(j)
                const VisNode l = vn[h.left];
                crossed = (v1x != h.cx || v1y != h.cy) && vis_cross(v1x, v1y, v2x, v2y, l.cx, l.cy, h.cx, h.cy);
            }
            if (!crossed && h.right >= 0) {                  // j -> right: h1 = centre(j), h2 = centre(right)
                const VisNode r = vn[h.right];
                crossed = (v1x != r.cx || v1y != r.cy) && vis_cross(v1x, v1y, v2x, v2y, h.cx, h.cy, r.cx, r.cy);
            }
        }
        if (crossed) mine[side == 0 ? 0 : 2] = -1;
    }
}

size_t scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)n, rocprim::plus<int32_t>(),
                                  (hipStream_t)0);
    return bytes;
}

}  // namespace

extern "C" int gte_knn_max_k(void) { return KNN_KMAX; }
extern "C" int gte_knn_max_page_nodes(void) { return KNN_PAGE_MAX; }

extern "C" int gte_knn_select(const int32_t* bbox, const int32_t* node_off, const int32_t* page_size, int64_t n_pages,
                              int64_t n_nodes, int64_t max_page_nodes, int k, int max_dist, int32_t* sel, void* stream) {
    if (n_pages <= 0 || n_nodes < 0 || k < 1 || k > KNN_KMAX || n_pages > 65535 || n_nodes >= INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "knn_select: bad sizes (1 <= k <= %d)", KNN_KMAX);
    if (max_page_nodes > KNN_PAGE_MAX)
        return gte::fail(GTE_ERR_UNSUPPORTED, "knn_select: a page of %lld boxes exceeds %d", (long long)max_page_nodes, KNN_PAGE_MAX);
    if (n_nodes == 0) return GTE_OK;
    if (!bbox || !node_off || !page_size || !sel) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "knn_select: null pointer");
    if (((uintptr_t)bbox & 15) != 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "knn_select: bbox must be 16-byte aligned");
    const size_t shm = (size_t)(max_page_nodes > 0 ? max_page_nodes : 1) * sizeof(Box);
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  KNN_PAGE_MAX * (int)sizeof(Box));
        configured = true;
    }
    hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)gte::ceil_div(max_page_nodes > 0 ? max_page_nodes : 1, 256), (unsigned)n_pages),
                       dim3(256), shm, gte::as_stream(stream), bbox, node_off, page_size, k, max_dist, sel);
    return gte::check_launch("knn_select");
}

// sel[n_nodes, 4]: top / right / bottom / left neighbour of every node (global ids, -1 none) after remove_vertical(); the graph
// is then gte_knn_csr(sel, k = 4, bidirectional = 1).
extern "C" int gte_visibility_select(const int32_t* bbox, const int32_t* node_off, const int32_t* page_size, int64_t n_pages,
                                     int64_t n_nodes, int64_t max_page_nodes, int max_dist, int32_t* sel, void* stream) {
    if (n_pages <= 0 || n_nodes < 0 || n_pages > 65535 || n_nodes >= INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "visibility_select: bad sizes");
    if (max_page_nodes > KNN_PAGE_MAX)
        return gte::fail(GTE_ERR_UNSUPPORTED, "visibility_select: a page of %lld boxes exceeds %d", (long long)max_page_nodes, KNN_PAGE_MAX);
    if (n_nodes == 0) return GTE_OK;
    if (!bbox || !node_off || !page_size || !sel) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "visibility_select: null pointer");
    if ((((uintptr_t)bbox | (uintptr_t)sel) & 15) != 0)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "visibility_select: bbox and sel must be 16-byte aligned");
    const size_t shm = (size_t)(max_page_nodes > 0 ? max_page_nodes : 1) * sizeof(Box);
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vis_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  KNN_PAGE_MAX * (int)sizeof(Box));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vis_prune_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  KNN_PAGE_MAX * (int)sizeof(VisNode));
        configured = true;
    }
    const dim3 grid((unsigned)gte::ceil_div(max_page_nodes > 0 ? max_page_nodes : 1, 256), (unsigned)n_pages);
    hipStream_t s = gte::as_stream(stream);
    hipLaunchKernelGGL(vis_select_kernel, grid, dim3(256), shm, s, bbox, node_off, page_size, max_dist, sel);
    hipLaunchKernelGGL(vis_prune_kernel, grid, dim3(256), shm, s, bbox, node_off, sel);
    return gte::check_launch("visibility_select");
}

extern "C" int64_t gte_knn_csr_workspace_bytes(int64_t n_nodes) {
    return gte::round_up((n_nodes + 1) * 4, 256) + (int64_t)gte::round_up((int64_t)scan_temp_bytes(n_nodes + 1), 256) + 256;
}

// in-edge CSR of the k-NN graph from the selection table: indptr[n+1] first (mode 0 + scan), the caller reads indptr[n] = E,
// provides indices / dst_of with >= E entries (E <= 2 k n), and calls again with fill = 1.
extern "C" int gte_knn_csr(const int32_t* sel, const int32_t* node_off, const int32_t* page_of_node, int64_t n_nodes, int k,
                           int bidirectional, int fill, int32_t* indptr, int32_t* indices, int32_t* dst_of, void* workspace,
                           int64_t workspace_bytes, void* stream) {
    if (n_nodes < 0 || k < 1 || k > KNN_KMAX || n_nodes >= INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "knn_csr: bad sizes");
    if (!indptr) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "knn_csr: indptr is NULL");
    hipStream_t s = gte::as_stream(stream);
    if (n_nodes == 0) return hipMemsetAsync(indptr, 0, 4, s) == hipSuccess ? GTE_OK : gte::fail(GTE_ERR_LAUNCH, "knn_csr: memset");
    if (!sel || !node_off || !page_of_node) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "knn_csr: null pointer");
    const dim3 grid((unsigned)gte::ceil_div(n_nodes, 256)), block(256);
    if (!fill) {
        if (!workspace || workspace_bytes < gte_knn_csr_workspace_bytes(n_nodes))
            return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "knn_csr: workspace too small");
        int32_t* deg = reinterpret_cast<int32_t*>(workspace);
        void* temp = reinterpret_cast<char*>(workspace) + gte::round_up((n_nodes + 1) * 4, 256);
        size_t temp_bytes = scan_temp_bytes(n_nodes + 1);
        if (hipMemsetAsync(deg + n_nodes, 0, 4, s) != hipSuccess) return gte::fail(GTE_ERR_LAUNCH, "knn_csr: memset");
        hipLaunchKernelGGL(knn_edges_kernel, grid, block, 0, s, sel, node_off, page_of_node, k, bidirectional, 0, deg,
                           (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int)n_nodes);
        hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, deg, indptr, 0, (size_t)(n_nodes + 1), rocprim::plus<int32_t>(), s);
        if (e != hipSuccess) return gte::fail(GTE_ERR_LAUNCH, "knn_csr: scan: %s", hipGetErrorString(e));
        return gte::check_launch("knn_csr degrees");
    }
    if (!indices || !dst_of) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "knn_csr: null output");
    hipLaunchKernelGGL(knn_edges_kernel, grid, block, 0, s, sel, node_off, page_of_node, k, bidirectional, 1, (int32_t*)nullptr,
                       indptr, indices, dst_of, (int)n_nodes);
    return gte::check_launch("knn_csr fill");
}

// island[v] = 1 for TEXT nodes from which no walk of exactly `khop` steps ends at a non-TEXT node (symmetric CSR).
// workspace: 2 * n bytes.
extern "C" int gte_island_mask(const int32_t* indptr, const int32_t* indices, const int32_t* label, int64_t n_nodes, int khop,
                               int text_label, uint8_t* island, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_nodes < 0 || khop < 0 || n_nodes >= INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "island_mask: bad sizes");
    if (n_nodes == 0) return GTE_OK;
    if (!indptr || !indices || !label || !island || !workspace) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "island_mask: null pointer");
    if (workspace_bytes < 2 * n_nodes) return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "island_mask: workspace too small");
    hipStream_t s = gte::as_stream(stream);
    uint8_t* cur = reinterpret_cast<uint8_t*>(workspace);
    uint8_t* nxt = cur + n_nodes;
    const dim3 grid((unsigned)gte::ceil_div(n_nodes, 256)), block(256);
    hipLaunchKernelGGL(island_init_kernel, grid, block, 0, s, label, text_label, cur, (int)n_nodes);
    for (int r = 0; r < khop; ++r) {
        hipLaunchKernelGGL(reach_round_kernel, grid, block, 0, s, indptr, indices, cur, nxt, (int)n_nodes);
        uint8_t* t = cur; cur = nxt; nxt = t;
    }
    hipLaunchKernelGGL(island_final_kernel, grid, block, 0, s, label, text_label, cur, island, (int)n_nodes);
    return gte::check_launch("island_mask");
}
