/* gte.h -- C ABI of libgte_hip.so: the MI355X (gfx950) hot path of GNN-TableExtraction.
 *
 * The reference (AILab-UniFI/GNN-TableExtraction) has no FFI: its boundary for this path
 * is a Python nn.Module API (src/components/graphs/models.py:15-116) that reaches the
 * arithmetic through DGL (update_all / gSpMM) and PyTorch (Linear, LayerNorm, ReLU,
 * CrossEntropyLoss, Adam).  Each entry point below replaces one of those call sites; the
 * reference file:line it replaces is cited on the declaration.  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no C++/torch types, no exceptions.
 *   - every pointer is a BORROWED DEVICE pointer (hipMalloc'ed by the caller, e.g. a torch
 *     tensor's data_ptr()); the caller keeps it alive until the stream has drained.
 *   - outputs and workspaces are pre-allocated by the caller; nothing is allocated inside
 *     (graph-capture safe).  Workspace sizes come from the *_workspace_bytes() queries.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every launch
 *     goes to that stream; no call synchronises.
 *   - return 0 on success, a negative gte_status otherwise; gte_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - re-entrant and thread-safe (autograd calls backward from another thread).
 *   - "ld*" arguments are leading dimensions in ELEMENTS (row stride); rows need only be
 *     4-byte aligned (F = 831 is a first-class shape), 16-byte alignment is faster.
 */
#ifndef GTE_H_
#define GTE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTE_VERSION 400 /* major*10000 + minor*100 + patch */

enum gte_status {
    GTE_OK = 0,
    GTE_ERR_INVALID_ARGUMENT = -1,
    GTE_ERR_LAUNCH = -2,
    GTE_ERR_WORKSPACE_TOO_SMALL = -3,
    GTE_ERR_UNSUPPORTED = -4,
};

enum gte_reduce { GTE_REDUCE_SUM = 0, GTE_REDUCE_MEAN = 1 }; /* MEAN: sum * 1/in_degree, 0 if none */
enum gte_dtype { GTE_F32 = 0, GTE_BF16 = 1 };

int gte_version(void);
const char* gte_last_error(void);
/* Device facts the host side sizes launches with: compute units, wave size, LDS bytes/CU. */
int gte_device_info(int* compute_units, int* wave_size, int* lds_bytes, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------
 * A6 / A9  neighbour aggregation (CSR gather SpMM)
 * replaces  models.py:53-54  g.update_all(fn.u_mul_e('h','feat','m'), fn.sum('m','h'))
 *           models.py:146-149 (fn.mean reducer)  and  models.py:74-78 get_norm (REDUCE_MEAN
 *           folds norm = 1/in_degree, inf->0, into the epilogue; in_degree = indptr[v+1]-indptr[v])
 *   out[v, 0:F] = scale_v * sum_{e in [indptr[v], indptr[v+1])} w[e] * x[indices[e], 0:F]
 * The same entry point run on the out-edge CSR is the backward of A6 (A7, DGL GSpMM.backward).
 * indptr: int32[n_rows+1]; indices: int32[nnz] (row ids of x); eweight: f32[nnz] or NULL (=1.0).
 * dtype GTE_BF16: x/out are bf16 (uint16 storage), accumulation in f32.
 * ---------------------------------------------------------------------------------------- */
int gte_spmm_csr(const int32_t* indptr, const int32_t* indices, const float* eweight,
                 const void* x, int64_t ldx, void* out, int64_t ldo,
                 int64_t n_rows, int64_t n_feat, int dtype, int reduce, void* stream);

/* The same contraction with the work split by EDGES instead of by rows -- the edge-parallel, wavefront-level segmented
 * reduction BASELINE.json's north_star names for the call site models.py:53-54: one wave per segment of 64 consecutive edges
 * of the CSR order, reduced segmented by destination row (lanes across the features, four source rows in flight); rows cut
 * by a segment boundary leave partial sums in `workspace` and a second pass adds them in segment order (deterministic).  For
 * graphs with hub rows (max in-degree > 64): a 3 000-edge row is 47 waves instead of one lane group walking 750 rounds.  fp32;
 * equal to gte_spmm_csr up to the summation order of rows that span segments. */
int64_t gte_spmm_csr_edge_workspace_bytes(int64_t n_edges, int64_t n_feat);
int gte_spmm_csr_edge(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* x, int64_t ldx,
                      float* out, int64_t ldo, int64_t n_rows, int64_t n_edges, int64_t n_feat, int reduce, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* Same contraction, but ACCUMULATING into out (out += ...): lets the backward add the
 * transpose-aggregated gradient onto the self-path gradient without a temporary. */
int gte_spmm_csr_accumulate(const int32_t* indptr, const int32_t* indices, const float* eweight,
                            const void* x, int64_t ldx, void* out, int64_t ldo,
                            int64_t n_rows, int64_t n_feat, int dtype, int reduce, void* stream);

/* LDS-staged variant of the same contraction for graphs with locality (page graphs: consecutive
 * destination rows share most sources).  A workgroup owns a tile of gte_spmm_tile_rows() consecutive
 * destination rows, stages the tile's DISTINCT source rows in LDS once per 32-float feature chunk and
 * reduces from LDS in CSR order (bit-identical to gte_spmm_csr).  Tile metadata is graph structure,
 * built once per graph by the caller:
 *   tile_ptr    int32[n_tiles+1]  offsets into tile_src        (n_tiles = ceil(n_rows / tile_rows))
 *   tile_src    int32[...]        distinct source rows of each tile
 *   local_index uint16[nnz]       position of indices[e] inside its tile's tile_src segment
 * Tiles with more than 128 distinct sources or 448 edges (or whose sources span 4 GB of x) are gathered directly (no size
 * limit on the graph).  Widths that are a multiple of 32 with at least 128 columns keep two chunks in flight (faster).
 * fp32 only.  accumulate != 0: out += ... */
int gte_spmm_tile_rows(void);
int gte_spmm_csr_tiled(const int32_t* indptr, const int32_t* indices, const uint16_t* local_index,
                       const float* eweight, const int32_t* tile_ptr, const int32_t* tile_src,
                       const float* x, int64_t ldx, float* out, int64_t ldo,
                       int64_t n_rows, int64_t n_feat, int reduce, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Graph preparation (what DGL does lazily per batched graph before gSpMM / its backward)
 * replaces  dgl.graph((u,v)) COO->CSC build (builder.py:425) and the reverse-CSR build DGL
 *           performs for GSpMM.backward; SURVEY 8(f) N1.
 * gte_coo_to_csr: stable counting sort of E edges by `key` (destination for the in-edge CSR,
 *   source for the out-edge CSR).  Writes indptr[n+1], indices[E] = other[perm], perm[E]
 *   (edge ids in row order; within a row ascending edge id => fixed summation order), and, if
 *   eweight != NULL, wout[E] = eweight[perm] * (scale_by_indeg ? 1/in_degree(dst) : 1).
 * workspace: gte_coo_to_csr_workspace_bytes(n, E).
 * ---------------------------------------------------------------------------------------- */
int64_t gte_coo_to_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int gte_coo_to_csr(const int32_t* key, const int32_t* other, const float* eweight,
                   const float* row_scale /* f32[n] or NULL: multiplies wout by row_scale[other] */,
                   int64_t n_nodes, int64_t n_edges,
                   int32_t* indptr, int32_t* indices, int32_t* perm, float* wout,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* Device-side batching of RESIDENT pages (replaces dgl.batch(...).to(device), model_train.py:297, per step).
 * Dataset arrays (built once): node_off[P+1], edge_off[P+1]; indptr_loc packed per page (page p's n_p+1 entries
 * start at node_off[p] + p and count from 0); indices_loc = column ids local to the page; weight in CSR order.
 * Batch = pages[n_batch] with b_node_off[n_batch+1], b_edge_off[n_batch+1] (exclusive prefix sums of the chosen
 * pages' node / CSR-entry counts, computed by the caller from host metadata).  Writes the block-diagonal CSR:
 * indptr_out[n_out+1], indices_out[e_out] (global ids), weight_out[e_out] (nullable).  Pure index work, no sort. */
int gte_batch_csr(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* edge_off,
                  const int32_t* b_node_off, const int32_t* b_edge_off, const int32_t* indptr_loc,
                  const int32_t* indices_loc, const float* weight, int32_t* indptr_out, int32_t* indices_out,
                  float* weight_out, int64_t n_out, int64_t e_out, void* stream);
/* ---- bf16 projection of the GAT (BASELINE configs[2] "4-head GAT bf16"; SURVEY A13: no reference symbol, parity unpinned) ------
 * gte_cast_bf16     y[r, :] = bf16(act(x[r, :])), act 0 = identity, 1 = ELU (the activation between GAT layers), zero padded to a
 *                   multiple of 8 columns; y 16-byte aligned, ldy a multiple of 8.
 * gte_gemm_bf16_nt  C[M, N] (fp32) = A[M, K] B[N, K]^T with bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation
 *                   (z = X W^T of a GAT layer; W stored [heads * dim, in_feats] like nn.Linear).  Operands 16-byte aligned, K and
 *                   the leading dimensions multiples of 8 (gte_cast_bf16 produces exactly that). */
int gte_cast_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int64_t rows, int64_t cols, int activation, void* stream);
int gte_gemm_bf16_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                     int64_t K, void* stream);

/* ---- k-NN page-graph construction --------------------------------------------------------------
 * Replaces GraphBuilder.get_graph(mode='knn') edge building (builder.py:240-292 over the projections of :383-395), dgl.to_simple +
 * dgl.to_bidirected (loader.py:319-320) and fast_remove_islands (builder.py:567-582) for a whole set of pages at once.
 * Inputs: bbox int32 [n_nodes, 4] (x0, y0, x1, y1; all pages concatenated, 16-byte aligned; boxes on the canvas:
 * 0 <= x0 <= x1 <= width, 0 <= y0 <= y1 <= height), node_off[n_pages + 1], page_size int32 [n_pages, 2] = (width, height).
 * Equal-distance candidates are ordered by (distance, node id) -- the reference's order among ties is an implementation detail
 * of CPython sets and numpy's argsort (oracle/knn_graph.py, pinned on the reference's own output, does the same).
 *   gte_knn_select  sel[n_nodes, k]: global ids of each node's selected neighbours (nearest first), -1 padded; k <= gte_knn_max_k(),
 *                   pages of at most gte_knn_max_page_nodes() boxes.
 *   gte_knn_csr     the in-edge CSR (rows = destinations, sources ascending, no duplicates) from sel: call with fill = 0 to get
 *                   indptr[n_nodes + 1] (workspace >= gte_knn_csr_workspace_bytes), read E = indptr[n_nodes], then with fill = 1 to
 *                   write indices[E] (sources) and dst_of[E] (the row of every entry: the COO form gte_edge_weights_bbox takes).
 *                   bidirectional != 0: symmetric closure (to_simple + to_bidirected); else the reference's directed edge list.
 *   gte_island_mask island[v] = 1 for nodes labelled text_label from which no walk of exactly khop steps over the (symmetric)
 *                   CSR ends at a node with another label; workspace >= 2 * n_nodes bytes.
 * Integer work only; no atomics, no sort: deterministic. */
/* Visibility mode of the same builder (builder.py:294-379): sel[n_nodes, 4] = every node's nearest visible box to the top /
 * right / bottom / left (global ids, -1 none; the reference's order-dependent update rules, restated in
 * oracle/visibility_graph.py and pinned on its own output) with the vertical edges that cross a horizontal edge removed
 * (remove_vertical()).  The graph is gte_knn_csr(sel, k = 4, bidirectional = 1).  bbox and sel 16-byte aligned. */
int gte_visibility_select(const int32_t* bbox, const int32_t* node_off, const int32_t* page_size, int64_t n_pages,
                          int64_t n_nodes, int64_t max_page_nodes, int max_dist, int32_t* sel, void* stream);
int gte_knn_max_k(void);
int gte_knn_max_page_nodes(void);
int gte_knn_select(const int32_t* bbox, const int32_t* node_off, const int32_t* page_size, int64_t n_pages, int64_t n_nodes,
                   int64_t max_page_nodes, int k, int max_dist, int32_t* sel, void* stream);
int64_t gte_knn_csr_workspace_bytes(int64_t n_nodes);
int gte_knn_csr(const int32_t* sel, const int32_t* node_off, const int32_t* page_of_node, int64_t n_nodes, int k,
                int bidirectional, int fill, int32_t* indptr, int32_t* indices, int32_t* dst_of, void* workspace,
                int64_t workspace_bytes, void* stream);
int gte_island_mask(const int32_t* indptr, const int32_t* indices, const int32_t* label, int64_t n_nodes, int khop,
                    int text_label, uint8_t* island, void* workspace, int64_t workspace_bytes, void* stream);

/* The whole batch in ONE launch (what the train loop calls every step; gte_batch_csr / gte_batch_rows are its pieces):
 * a page's rows are CONTIGUOUS in the resident arrays and in the batch, so every array of the batch is the concatenation
 * of per-page runs -- features and labels copied, indptr / indices copied with a per-page constant added.  Workgroup
 * (page i, worker w) moves its share of each of page i's runs with 16-byte accesses (rows need only 4-byte alignment);
 * no per-element search, no allocation, no synchronisation.  HBM-bound: 2 * n_out * n_cols * 4 bytes dominate.
 * `in` / `out`: the two CSR directions (in-edge, out-edge); any weight pointer may be NULL (then the pair is skipped);
 * label / label_out may be NULL.  feat_out has leading dimension n_cols (packed). */
typedef struct gte_batch_arrays {
    const int32_t* edge_off;      /* [P+1]  resident: first CSR entry of every page                                  */
    const int32_t* indptr_loc;    /* packed per-page indptr (see gte_batch_csr)                                        */
    const int32_t* indices_loc;   /* page-local column ids                                                             */
    const float* weight;          /* CSR-ordered weights or NULL                                                       */
    const int32_t* b_edge_off;    /* [n_batch+1] batch: first entry of every chosen page                               */
    int32_t* indptr_out;          /* [n_out+1]                                                                         */
    int32_t* indices_out;         /* [e_out]                                                                           */
    float* weight_out;            /* [e_out] or NULL                                                                   */
} gte_batch_arrays;
int gte_batch_assemble(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* b_node_off,
                       const gte_batch_arrays* in_edges, const gte_batch_arrays* out_edges,
                       const float* feat, int64_t ld_feat, int64_t n_cols, float* feat_out,
                       const float* label, float* label_out, int64_t n_out, void* stream);
/* 1: a gte_batch_assemble / gte_batch_assemble_rows call made while a fold deferral is open on its stream (i.e. from inside a
 * training step: gte_gcnsage_step's before-last-GEMM callback) is not launched; the deferral's flush -- the fold + optimiser launch
 * that ends the step -- carries it as extra workgroups (the NEXT batch's buffers are independent of everything that launch
 * touches).  One job per deferral; without an open deferral on that stream the call launches as always.  0 (default): off.
 * Per host thread; returns the previous setting. */
int gte_batch_assemble_defer(int on);
/* out[b_node_off[i] + r, 0:n_cols] = in[node_off[pages[i]] + r, 0:n_cols]  (features, labels stored as f32) */
int gte_batch_rows(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* b_node_off,
                   const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t n_out, int64_t n_cols,
                   void* stream);
/* Edge weights of loader.py:332-344: w_e = 1 - d_e / max_{e in the page} d_e with d = the reference's integer box
 * distance (graphs/utils.py:56-88).  bbox int32[N,4] (x0,y0,x1,y1; 16-byte aligned), graph_of_node int32[N].
 * A page whose distances are all 0 gets weight 1 (the reference divides by zero there).  Bit-exact vs the
 * double-precision host formula. */
int64_t gte_edge_weights_workspace_bytes(int64_t n_edges, int64_t n_graphs);
int gte_edge_weights_bbox(const int32_t* bbox, const int32_t* src, const int32_t* dst, const int32_t* graph_of_node,
                          int64_t n_edges, int64_t n_graphs, float* weight, void* workspace, int64_t workspace_bytes,
                          void* stream);
/* BBOX node features (replaces nlp/bbox.py:49-124 called at model_train.py:293): per word
 * out[i, 0:13] = [w, h, cx, cy, w*h, x0, y0, x1, y1, hist_letters, hist_digits, hist_others, hist_empty].
 * bbox int32[N,4]; char_counts int32[N,3] = (#letters, #digits, #other characters) of the word, counted on the host;
 * float64 arithmetic as the reference, stored as float32.  Bit-exact vs the host formula. */
int gte_bbox_features(const int32_t* bbox, const int32_t* char_counts, float* out, int64_t ldo, int64_t n_nodes,
                      void* stream);
/* inv_deg[v] = 1/(indptr[v+1]-indptr[v]) or 0  (models.py:74-78 as a standalone vector) */
int gte_inv_degree(const int32_t* indptr, float* inv_deg, int64_t n_nodes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A8  per-node transform: split-weight linear + bias + LayerNorm + ReLU, fp32 MFMA
 * replaces  models.py:69-72 (torch.cat((h, ah*norm),1)), :63 nn.Linear(2F,out), :64 nn.LayerNorm,
 *           :65-66 activation.
 *   z = a1[M,k1] * W[:, 0:k1]^T + a2[M,k2] * W[:, k1:k1+k2]^T + bias        (W: [n_out, k1+k2])
 *   y = relu?( LN?(z) ) ; LN over n_out with eps, affine (gamma, beta)
 * a2 may be NULL (k2 = 0; use_pp=True path, models.py:49).  gamma == NULL => no LayerNorm.
 * z_save (nullable): pre-LayerNorm z for the backward; stats (nullable): f32[2*M] = mean, rstd.
 * ---------------------------------------------------------------------------------------- */
int gte_sage_linear_fwd(const float* a1, int64_t lda1, int64_t k1,
                        const float* a2, int64_t lda2, int64_t k2,
                        const float* W, int64_t ldw, const float* bias,
                        const float* gamma, const float* beta, float eps, int relu,
                        float* z_save, int64_t ldz, float* stats,
                        float* y, int64_t ldy, int64_t M, int64_t n_out, void* stream);
/* 1 when gte_sage_linear_fwd runs this shape as ONE pass (linear + LayerNorm + ReLU: k1 + k2 <= 64, n_out % 4 == 0,
 * n_out <= 256 -- the BBOX-only input layer, F0 = 13), 0 when it is an MFMA GEMM followed by the LayerNorm launch. */
int gte_sage_linear_fwd_fuses_ln(int64_t k_total, int64_t n_out);

/* Weight gradient of the split-weight linear, both halves in ONE launch:
 *   dW[n_out, k1+k2] = dZ[M, n_out]^T * [x1[M,k1] | x2[M,k2]]        (autograd of models.py:63 w.r.t. the weight)
 * The reduction runs over the M nodes (split over workgroups, deterministic slab reduction).  x2 may be
 * NULL (k2 = 0).  workspace: gte_sage_linear_dw_workspace_bytes(n_out, k1, k2, M). */
int64_t gte_sage_linear_dw_workspace_bytes(int64_t n_out, int64_t k1, int64_t k2, int64_t n_nodes);
int gte_sage_linear_dw(const float* dz, int64_t lddz, const float* x1, int64_t ldx1, int64_t k1,
                       const float* x2, int64_t ldx2, int64_t k2, float* dW, int64_t lddw,
                       int64_t n_out, int64_t n_nodes, void* workspace, int64_t workspace_bytes, void* stream);

/* gte_spmm_csr_accumulate with a LayerNorm(+ReLU) epilogue, for a layer in transform-then-aggregate order
 * (models.py:53-54 then :64-66): z[v,:] += scale_v * sum_e w[e] x[src,:] (z written back: the LayerNorm backward
 * needs it), then y[v,:] = relu?(gamma * (z - mean) / sqrt(var + eps) + beta) and stats[v] = mean,
 * stats[n_rows + v] = rstd (nullable) -- same statistics as gte_ln_relu_fwd.  f32 only, n_feat <= 1024 (a row lives in
 * the registers of one lane group; gte_spmm_csr_accumulate_ln_supported says so).  The kernel works on 16-byte chunks:
 * when n_feat % 4 != 0 (% 16 != 0 with a P3 image of y) the rows of x, z and y must be PADDED -- allocated to the next
 * multiple of 4 (16) floats, ld >= that; the padding of x is read and must hold zeros, the padding of z / y / the image is
 * written as zeros; mean and variance run over the n_feat true columns (models.py:64). */
int gte_spmm_csr_accumulate_ln_supported(int64_t n_feat);
int gte_spmm_csr_accumulate_ln(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* x,
                               int64_t ldx, float* z, int64_t ldz, int64_t n_rows, int64_t n_feat, int reduce,
                               const float* gamma, const float* beta, float eps, int relu, float* y, int64_t ldy,
                               float* stats, void* stream);

/* ---- GEMM tail split ----------------------------------------------------------------------------------------------
 * A whole-K GEMM launch (gte_sage_linear_fwd, gte_sage_transform_fwd, gte_sage_qform_dx, gte_gemm_f32 without split-K)
 * whose last round of tiles would leave more than half of the compute units idle cuts those tiles' reduction range into
 * pieces that fill the idle units, and finishes them with a small fix-up launch.  It needs scratch for the partial
 * tiles: gte_gemm_set_tail_workspace registers a caller-owned device buffer (>= gte_gemm_tail_workspace_bytes()) for the
 * calling host thread; NULL / 0 unregisters (launches then never split).  Launches that may split must not run
 * concurrently on two streams with one workspace.  Results are deterministic either way (fixed summation order), but a
 * split tile sums its K range in pieces: last-bit differences against the unsplit launch. */
int64_t gte_gemm_tail_workspace_bytes(void);
int gte_gemm_set_tail_workspace(void* workspace, int64_t workspace_bytes);

/* ---- GEMM arithmetic mode ------------------------------------------------------------------------------------------
 * How the fp32 transform GEMMs (every entry point of this section and gte_gemm_f32) multiply.  Process-wide.
 *   GTE_GEMM_F32        v_mfma_f32_32x32x2_f32: bit-exact fp32 FMA chains.
 *   GTE_GEMM_SPLIT_BF16 (default since round 3) every fp32 operand is cut exactly into three bf16 pieces (24 significand bits) in the kernel and the
 *                       product is six v_mfma_f32_32x32x16_bf16 partial products accumulated in fp32 (csrc/gemm_split.h):
 *                       fp32 in, fp32 out, error against fp64 not above the fp32 kernel's, 2.67x its matrix-pipe rate.
 *                       Tiles narrower than 128 columns and the small-shape path stay on the fp32 kernel.  Non-finite
 *                       operands give NaN where the fp32 kernel returns inf.
 * The environment variable GTE_GEMM_MODE=f32 selects the fp32 MFMA kernel before the first call; gte_gemm_set_mode overrides. */
#define GTE_GEMM_F32 0
#define GTE_GEMM_SPLIT_BF16 1
int gte_gemm_set_mode(int mode);
int gte_gemm_get_mode(void);
/* Override for the calling host thread only (-1: none, the process-wide mode applies): one caller's GEMMs in another mode
 * without touching what other threads get.  gte_gemm_get_mode reports the mode the calling thread would run in. */
int gte_gemm_set_thread_mode(int mode);

/* ---- planes GEMMs (P3 operands) --------------------------------------------------------------------------------------
 * The same transform GEMMs (models.py:63 nn.Linear and its autograd) on operands their PRODUCER has already cut into the
 * three bf16 planes of the split mode (csrc/p3.h): the GEMM then moves bf16 planes memory -> LDS (LDS-DMA) and spends its
 * issue slots on the six v_mfma_f32_32x32x16_bf16 products per fragment pair only.  Results are bit-identical to the split
 * mode on the fp32 operand (same pieces, same products, same order).
 * P3 image of a logical fp32 matrix [rows][cols]: row r at byte r * ldp (ldp >= gte_p3_row_bytes(cols), a multiple of 16; the
 * image 16-byte aligned); 16-feature block fb at + 96 fb; in a block plane h (16 bf16), plane m, plane l; x = h + m + l
 * exactly; features >= cols are zero.  "ldp*" arguments of this section are row strides in BYTES.
 * BLOCK-MAJOR images (round 6; the weights -- the B operand "b, ldb" of the gte_gemm_p3_nt* entry points -- and every gte_p3_desc /
 * gte_p3_from_f32 / gte_p3_to_f32 image): a NEGATIVE ldp names an image whose 16-feature block fb is ONE contiguous run at byte
 * fb * (-ldp), row r at + 96 r inside it (-ldp >= 96 rows, a multiple of 16).  A K block of the weights is then whole cache lines
 * (row-major: 96-byte runs at the row stride, every line fetched twice), and where a product has one column of 256-wide tiles the
 * block-major-weights kernel (csrc/gemm_p3.hip gemm_p3_nt_sq_kernel) loads its weight fragments straight into registers.  Same
 * products, same bits as the row-major image.  A operands, row-mapped resident images and the TN operands stay row-major. */
int64_t gte_p3_row_bytes(int64_t cols);
/* dst(r, c) = transpose ? src[c * ld + r] : src[r * ld + c]  for r < rows, c < cols (ld in elements) */
int gte_p3_from_f32(const float* src, int64_t ld, int64_t rows, int64_t cols, int transpose, void* dst, int64_t ldp,
                    void* stream);
int gte_p3_to_f32(const void* src, int64_t ldp, int64_t rows, int64_t cols, float* dst, int64_t ld, void* stream);
/* up to 16 of them in ONE launch (the weight images of every layer, after an optimiser step) */
typedef struct gte_p3_desc {
    const float* src; int64_t ld;      /* fp32 source, leading dimension in elements */
    int64_t rows, cols;                /* of the IMAGE: dst(r, c) = transpose ? src[c * ld + r] : src[r * ld + c] */
    int transpose;
    void* dst; int64_t ldp;            /* image (first byte of its row 0 / column block 0), row stride in bytes (< 0: block-major) */
} gte_p3_desc;
int gte_p3_from_f32_batch(const gte_p3_desc* descs, int n, void* stream);
/* Producers that write their result as a P3 image: the aggregation (q = A_w^T (norm dz), aggregated inputs; any n_feat, the
 * image's columns up to the next multiple of 16 are written as zeros), the fused aggregation + LayerNorm(+ReLU) (y as image
 * and / or fp32: either may be NULL) and the LayerNorm backward (dz as fp32 AND image: fp32 feeds the transpose aggregation,
 * the image the dX / dW GEMMs; widths other than 128 <= n_out <= 512 with n_out % 16 == 0 run a masked kernel on PADDED
 * rows: LayerNorm required, n_out <= 1024, lddy / ldz / lddz >= n_out rounded up to 4).  Same arithmetic as their fp32 forms
 * (gte_spmm_csr, gte_spmm_csr_accumulate_ln, gte_ln_relu_bwd); the image holds exactly the fp32 values. */
int gte_sage_linear_fwd_p3(const float* a1, int64_t lda1, int64_t k1, const float* a2, int64_t lda2, int64_t k2, const float* W,
                           int64_t ldw, const float* bias, const float* gamma, const float* beta, float eps, int relu,
                           float* z_save, int64_t ldz, float* stats, float* y /* nullable */, int64_t ldy, void* yp3,
                           int64_t ldyp3, int64_t M, int64_t n_out, void* stream);   /* one-pass form only (fuses_ln) */
/* Backward of the short-input layer (the layer gte_sage_linear_fwd runs in one pass: k1 + k2 <= 28 here) when it is the INPUT
 * layer (no dX): LayerNorm(+ReLU) backward and dW = dz^T [a1 | a2] in ONE pass over dy.  z is recomputed from the 26 inputs
 * per row with the forward kernel's instruction sequence (bit-identical: the forward need not save z -- pass z_save = NULL
 * there), dz never reaches memory.  Replaces gte_ln_relu_bwd + gte_sage_linear_dw (models.py:63-66 autograd).
 * dgamma / dbeta / dbias nullable.  The partial sums join an open fold deferral. */
int gte_sage_smallk_bwd_supported(int64_t k_total, int64_t n_out);
int64_t gte_sage_smallk_bwd_workspace_bytes(int64_t n_nodes, int64_t k_total, int64_t n_out);
int gte_sage_smallk_bwd(const float* dy, int64_t lddy, const float* a1, int64_t lda1, int64_t k1, const float* a2, int64_t lda2,
                        int64_t k2, const float* W, int64_t ldw, const float* bias, const float* gamma, const float* beta,
                        const float* stats, int relu, float* dW, int64_t lddw, float* dbias, float* dgamma, float* dbeta,
                        int64_t n_nodes, int64_t n_out, void* workspace, int64_t workspace_bytes, void* stream);
int gte_spmm_csr_p3(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* x, int64_t ldx,
                    void* outp3, int64_t ldp, int64_t n_rows, int64_t n_feat, int reduce, void* stream);
int gte_spmm_csr_accumulate_ln_p3(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* x,
                                  int64_t ldx, float* z, int64_t ldz, int64_t n_rows, int64_t n_feat, int reduce,
                                  const float* gamma, const float* beta, float eps, int relu, float* y, int64_t ldy,
                                  void* yp3, int64_t ldyp3, float* stats, void* stream);
int gte_ln_relu_bwd_p3(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* stats, const float* gamma,
                       const float* beta, int relu, float* dz, int64_t lddz, void* dzp3, int64_t ldp3, float* dgamma,
                       float* dbeta, float* dbias, int64_t M, int64_t n_out, void* workspace, int64_t workspace_bytes,
                       void* stream);
/* c[m, n] (+)= [a1 | a2] b^T (+ bias on columns < bias_cols; <= 0: all), optional relu.  a1: P3 [m][k1]; a2: P3 [m][k2]
 * (NULL, k2 = 0: one K segment); b: P3 [n][...] whose rows hold the ceil(k1 / 16) blocks of segment 1 followed by the blocks
 * of segment 2.  replaces models.py:69-72 + :63 (forward; transform-first layers use b = [W_s ; W_n] as 2 out rows) and
 * the dX product of its backward (a1 = dz, a2 = q, b = [W_s^T | W_n^T]). */
int gte_gemm_p3_nt(const void* a1, int64_t ldpa1, int64_t k1, const void* a2, int64_t ldpa2, int64_t k2, const void* b,
                   int64_t ldpb, const float* bias, int64_t bias_cols, float* c, int64_t ldc, int64_t m, int64_t n, int relu,
                   int accumulate, void* stream);
/* c[m, n] = a^T b over k rows (dW = dZ^T X; split over the rows, partial slabs in `workspace`, folded in a fixed order --
 * inside an open fold deferral by gte_fold_defer_flush).  a: P3 [k][m], b: P3 [k][n].  nseg > 0: two column segments of
 * nseg columns (n == 2 nseg): c[:, 0:nseg] = a^T b, c[:, nseg:] = a2^T b2 (a2 / b2 NULL: the operand of segment 0). */
/* ... with A = the rows a_rows[0 .. m) of a RESIDENT image a_res [n_res_rows][k] (one K segment): the input layer's forward
 * transform straight from the resident features -- no per-batch copy of the rows (gte_batch_assemble_rows writes the map).
 * Images below 4 GB are read through 32-bit buffer offsets, larger ones through 64-bit per-lane addresses (same results). */
int gte_gemm_p3_nt_rows(const void* a_res, int64_t ldpa, int64_t k, const int32_t* a_rows, int64_t n_res_rows, const void* b,
                        int64_t ldpb, const float* bias, int64_t bias_cols, float* c, int64_t ldc, int64_t m, int64_t n,
                        int relu, int accumulate, void* stream);
/* ... with BOTH K segments resident: [A | A2] = the rows a_rows[0 .. m) of two resident images of n_res_rows rows and k columns
 * each -- the input features and their CACHED mean aggregate norm . A_w x (page-local and constant over a run: made once when the
 * pages are loaded).  The forward of models.py:69-72 `linear(cat(h, ah * norm))` for the input layer without any per-batch
 * operand preparation: no aggregation launch, no image conversion, no feature copy.  b = P3 [n][2 ceil16(k)], the W_n columns at
 * block ceil(k / 16).  Both images below 4 GB (32-bit row offsets). */
int gte_gemm_p3_nt_rows2(const void* a_res, int64_t ldpa, const void* a2_res, int64_t ldpa2, int64_t k, const int32_t* a_rows,
                         int64_t n_res_rows, const void* b, int64_t ldpb, const float* bias, int64_t bias_cols, float* c, int64_t ldc,
                         int64_t m, int64_t n, int relu, int accumulate, void* stream);
/* gte_gemm_p3_nt + gte_ln_relu_fwd_p3 in ONE launch (n <= 256: a workgroup's tile holds whole rows): z = [a1 | a2] b^T + bias is
 * written as fp32 (ldz >= n rounded up to 4: the operand of the layer's LayerNorm backward) and normalised by the workgroup that
 * computed it -- models.py:63-66 `linear`, `lynorm`, `activation` of an aggregate-first layer: stats = {mean [m], rstd [m]}, y =
 * relu?(LN(z)) as fp32 (nullable) and as a P3 image (nullable; columns up to the next multiple of 16 zero).  Bit-identical to the two
 * launches.  ..._rows2_...: [a1 | a2] = mapped rows of two resident images (gte_gemm_p3_nt_rows2). */
int gte_gemm_p3_nt_ln_fwd_supported(int64_t n);
int gte_gemm_p3_nt_ln_fwd(const void* a1, int64_t ldpa1, int64_t k1, const void* a2, int64_t ldpa2, int64_t k2, const void* b,
                          int64_t ldpb, const float* bias, const float* gamma, const float* beta, float eps, int relu, float* z,
                          int64_t ldz, float* y, int64_t ldy, void* yp3, int64_t ldyp3, float* stats, int64_t m, int64_t n,
                          void* stream);
int gte_gemm_p3_nt_rows2_ln_fwd(const void* a_res, int64_t ldpa, const void* a2_res, int64_t ldpa2, int64_t k, const int32_t* a_rows,
                                int64_t n_res_rows, const void* b, int64_t ldpb, const float* bias, const float* gamma,
                                const float* beta, float eps, int relu, float* z, int64_t ldz, float* y, int64_t ldy, void* yp3,
                                int64_t ldyp3, float* stats, int64_t m, int64_t n, void* stream);
/* gte_gemm_p3_nt (no bias / relu / accumulate) whose product dy [m][n], n <= 256, is NOT stored: the workgroup that computed a
 * block of rows runs the LayerNorm(+ReLU) backward of those rows on it (models.py:64-66 autograd of the layer below): dz as fp32
 * and as a P3 image (dzp3 nullable), column sums dgamma / dbeta / dbias (nullable) into the fold deferral.  z / stats / gamma /
 * beta as gte_ln_relu_bwd.  Replaces gte_gemm_p3_nt + gte_ln_relu_bwd_p3 (one launch, 2 m n 4 bytes of traffic less); dz is
 * bit-identical to theirs.  Any n in 1 .. 256; n % 4 != 0 needs the rows of z and dz padded to a multiple of 4 floats (ldz,
 * lddz >= that, padding of dz written as zeros), and whenever n % 16 != 0 the image's columns up to the next multiple of 16 are
 * written as zeros.  dz is nullable when dzp3 is given (a layer below whose weight gradient reads the image only). */
int gte_gemm_p3_nt_ln_bwd_supported(int64_t n);
int64_t gte_gemm_p3_nt_ln_bwd_workspace_bytes(int64_t m, int64_t n);
int gte_gemm_p3_nt_ln_bwd(const void* a1, int64_t ldpa1, int64_t k1, const void* a2, int64_t ldpa2, int64_t k2, const void* b,
                          int64_t ldpb, const float* z, int64_t ldz, const float* stats, const float* gamma, const float* beta,
                          int relu, float* dz, int64_t lddz, void* dzp3, int64_t ldp3, float* dgamma, float* dbeta, float* dbias,
                          int64_t m, int64_t n, void* workspace, int64_t workspace_bytes, void* stream);
/* ... with the WHOLE backward of a short-input INPUT layer below (gte_sage_smallk_bwd: k1 + k2 <= 28) as the epilogue: dy is not
 * stored, z is recomputed per row, dW / dbias / dgamma / dbeta of that layer come out through the fold deferral (outside a
 * deferral dW must be packed, lddw == k1 + k2).  Replaces gte_gemm_p3_nt + gte_sage_smallk_bwd. */
int gte_gemm_p3_nt_smallk_bwd_supported(int64_t k_total, int64_t n);
int64_t gte_gemm_p3_nt_smallk_bwd_workspace_bytes(int64_t m, int64_t k_total, int64_t n);
int gte_gemm_p3_nt_smallk_bwd(const void* a1, int64_t ldpa1, int64_t kg1, const void* a2, int64_t ldpa2, int64_t kg2, const void* b,
                              int64_t ldpb, const float* x, int64_t ldx, int64_t k1, const float* ahn, int64_t ldahn, int64_t k2,
                              const float* W, int64_t ldw, const float* bias, const float* gamma, const float* beta,
                              const float* stats, int relu, float* dW, int64_t lddw, float* dbias, float* dgamma, float* dbeta,
                              int64_t m, int64_t n, void* workspace, int64_t workspace_bytes, void* stream);
int64_t gte_gemm_p3_tn_workspace_bytes(int64_t m, int64_t n, int64_t nseg, int64_t k);
int gte_gemm_p3_tn(const void* a, int64_t ldpa, const void* a2, int64_t ldpa2, const void* b, int64_t ldpb, const void* b2,
                   int64_t ldpb2, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n, int64_t k, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* gte_gemm_p3_tn with b = the rows b_rows[0 .. k) of a RESIDENT image b_res (the input layer's dW).  b_rows holds k rounded up
 * to 16, plus 1, entries; the entries past k = n_res_rows (a row past the image reads as zeros).  An image of 4 GB or more is
 * read through 64-bit addresses WITHOUT a range check: its allocation must then hold row n_res_rows, filled with zeros. */
int gte_gemm_p3_tn_rows(const void* a, int64_t ldpa, const void* a2, int64_t ldpa2, const void* b_res, int64_t ldpb,
                        const int32_t* b_rows, int64_t n_res_rows, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n,
                        int64_t k, void* workspace, int64_t workspace_bytes, void* stream);
/* ... with two resident images behind the map: c[:, 0:nseg] = a^T b_res[rows], c[:, nseg:] = a^T b2_res[rows] (n == 2 nseg; same
 * row stride; the zero-row requirement above holds for both): dW = [dz^T x | dz^T ahn] of the input layer with a cached aggregate
 * (autograd of models.py:63 for `cat(h, ah * norm)`) -- the q = A_w^T (norm dz) aggregation of layer 0 is not needed at all. */
int gte_gemm_p3_tn_rows2(const void* a, int64_t ldpa, const void* b_res, int64_t ldpb, const void* b2_res, int64_t ldpb2,
                         const int32_t* b_rows, int64_t n_res_rows, int64_t nseg, float* c, int64_t ldc, int64_t m, int64_t n,
                         int64_t k, void* workspace, int64_t workspace_bytes, void* stream);
/* Second half of the fused head (gte_head_agg_ce) when the output layer's products run on the planes GEMMs: dlq [n][>= 32] holds
 * dl in columns 0 .. C-1 WITHOUT the 1 / sum(w) of the weighted cross-entropy (model_train.py:171,327) and -- when rindptr is
 * NULL -- q = A_w^T (norm dl) in columns 16 .. 16 + C-1; with the out-edge CSR (rindptr / rindices / w_out = w / in_degree(dst))
 * the launch forms q itself (autograd of models.py:53-54 for the class-count-wide aggregation).  It folds the CE partials of
 * gte_head_agg_ce (ce_partial), publishes out3 = {loss, sum w, #correct}, writes alpha [dl | q], alpha = grad_scale / sum w, as
 * ONE P3 image [n][32] (the operand of the layer's dW / dh GEMMs) and gbias [C] = column sums of alpha dl (block partials in
 * `workspace`, folded in order -- inside an open fold deferral by gte_fold_defer_flush). */
int64_t gte_head_dlq_finish_workspace_bytes(int64_t n_nodes);
int gte_head_dlq_finish(const int32_t* rindptr, const int32_t* rindices, const float* w_out, const float* dlq, int64_t lddlq,
                        int64_t n_nodes, int64_t n_classes, const void* ce_partial, float grad_scale, float* out3, void* dlqp3,
                        int64_t ldp, float* gbias, void* workspace, int64_t workspace_bytes, void* stream);
/* Which kernel family gte_gemm_p3_nt / _rows / _rows2 (epilogue 0), gte_gemm_p3_nt_ln_bwd (1; 3 = a width that is not a multiple
 * of 16) or gte_gemm_p3_nt[_rows2]_ln_fwd (4) would launch for [m x n] = [a1 | a2] b^T with k1 + k2 columns: 1 = the
 * block-major-weights kernel (weight fragments straight into registers, A through 64-deep LDS slots; needs block-major weights, one
 * column of 256-wide tiles, at least 32 K blocks, K padded by at most 1 / 12), 0 = the loader-wave / ring kernels.  *row_tile = the row
 * tile of the launch where it is chosen by the one-round rule (32 / 64 / 96, else 128), 0 where the tile chooser picks.  Host logic
 * only: with cus > 0 no device is touched (tests run it on the CPU); cus <= 0 reads the device's CU count. */
int gte_gemm_p3_nt_plan(int64_t m, int64_t n, int64_t k1, int64_t k2, int weights_block_major, int epilogue, int cus, int* row_tile);
/* The three setters below are TEST HOOKS with THREAD-LOCAL effect: they change what the calling host thread's later launches pick,
 * nothing any other thread sees (the library keeps no mutable process-wide state besides the GEMM mode above).  Environment
 * variables: the shipped library reads GTE_GEMM_MODE only; every other switch of earlier rounds lives in the measurement build
 * (libgte_hip_measure.so, -DGTE_MEASURE: csrc/gte_common.h).
 * Tile configuration of gte_gemm_p3_nt / _rows / _rows2: -1 = chosen per problem (default), 0 ... 7 = forced (tests run every
 * configuration against the same bits; 0 - 6 the 128- and 256-column tiles, 7 a measurement tile). */
int gte_gemm_p3_set_nt_cfg(int cfg);
/* Row tile of the launches with a LayerNorm epilogue (gte_gemm_p3_nt_ln_fwd / _rows2_ln_fwd / _ln_bwd; images below 4 GB): 0 = the
 * smallest of 32 / 64 / 96 rows that covers m in one round of workgroups, else 128 (default); 32 / 64 / 96 / 128 = forced (tests run
 * every tile against the same bits). */
int gte_gemm_p3_set_ln_rows(int rows);
/* Row maps: 0 = 64-bit addresses for resident images of 4 GB or more only (default), 1 = always (tests, A/B timing; the
 * zero-row requirement of gte_gemm_p3_tn_rows then holds for every image). */
int gte_gemm_p3_set_rows64(int mode);
/* gte_batch_assemble with a ROW MAP instead of (feat == NULL) or next to the copied feature rows: row_map[r] = resident row of
 * batch row r (r < n_out), row_map[n_out .. n_out + row_map_pad) = n_res_rows. */
int gte_batch_assemble_rows(const int32_t* pages, int64_t n_batch, const int32_t* node_off, const int32_t* b_node_off,
                            const gte_batch_arrays* in_edges, const gte_batch_arrays* out_edges, const float* feat,
                            int64_t ld_feat, int64_t n_cols, float* feat_out, const float* label, float* label_out,
                            int64_t n_out, int32_t* row_map, int64_t row_map_pad, int64_t n_res_rows, void* stream);

/* ---- one optimisation step from a prepared plan -------------------------------------------------------------------------
 * replaces the batch loop body of model_train.py:320-332 (logits = model(g); loss; zero_grad; backward; optimizer.step())
 * for the configuration every shipped run of the reference uses -- GcnSAGE with ReLU + LayerNorm hidden layers, dropout 0,
 * a class-count-wide output layer -- as ONE host call: the launches of a step (see models/engine.py for the same sequence
 * issued call by call) are issued from here, so the host cost of a step no longer grows with its ~17 launches (the
 * BBOX-only configurations, F0 = 13, were host-bound at ~0.25 ms of ctypes calls per 0.37 ms step).
 * The plan only borrows device pointers; nothing is allocated or synchronised.  Layer kinds:
 *   GTE_LAYER_PLANES  t = h [W_s ; W_n]^T on the planes GEMM, z = t_self + b + mean-aggregate(t_neigh), LayerNorm, ReLU;
 *                     input / dz / q as P3 images.  Any fout <= 1024 (the reference's runs: --h_layer_dim=1000 or
 *                     int(calculate_hidden) = 96 ... 218, run_multiple_train.sh:8-113): the fp32 row buffers of the layer are
 *                     PADDED to ldf = fout rounded up to 16 floats per row (t: two halves of ldf), the weight image holds zero
 *                     rows behind the fout rows of each half, LayerNorm statistics run over the true width
 *   GTE_LAYER_SMALLK  input layer with k = 2 fin <= 64 (BBOX features): aggregate-first, linear + LayerNorm + ReLU in one pass
 *   GTE_LAYER_AGGFIRST input layer that widens (fin < fout, e.g. 13 / 63 / 363 -> 1000): x and mean-aggregate(x) as P3 images,
 *                     z = [x | ahn] W^T + b on the planes GEMM (two K segments), LayerNorm + ReLU (gte_ln_relu_fwd_p3);
 *                     dW = [dz^T x | dz^T ahn] on the TN planes GEMM
 * followed by the output layer: the narrow kernels (gte_sage_narrow_*, gte_head_agg_ce; out_fin <= 256, out_fin % 8 == 0) or,
 * out_gemm = 1, the planes GEMMs (N = 32 forward; dW_out with M = n_classes; dh with K = 32).  phase: 0 the whole step; 1
 * everything up to the last weight-gradient GEMM, 2 the rest (the train loop queues the NEXT batch's assembly on its side
 * stream in between).
 *   GTE_LAYER_CACHED  input layer whose input AND its mean aggregate are RESIDENT P3 images read through the batch's row map
 *                     (hp / ahnp + h_rows): z = [x | ahn] W^T + b (gte_gemm_p3_nt_rows2), LayerNorm + ReLU; dW = [dz^T x | dz^T ahn]
 *                     (gte_gemm_p3_tn_rows2).  No aggregation, no q, no copy of the input in the step.  Any fin >= 16, fout <= 1024
 */
enum gte_layer_kind { GTE_LAYER_PLANES = 0, GTE_LAYER_SMALLK = 1, GTE_LAYER_AGGFIRST = 2, GTE_LAYER_CACHED = 3 };
typedef struct gte_step_layer {
    int kind;
    int64_t fin, fout;
    const float* W; const float* bias; const float* gamma; const float* beta; float eps; int relu;
    float* gW; float* gbias; float* ggamma; float* gbeta;
    void* wimg_fwd; int64_t ldp_wfwd;      /* PLANES: P3 [2 fout][fin]   = [W_s rows ; W_n rows]            */
    void* wimg_bwd; int64_t ldp_wbwd;      /* PLANES, layer > 0: P3 [fin][2 fout] = [W_s^T | W_n^T]         */
    const float* x; int64_t ldx;           /* fp32 input rows (SMALLK; PLANES layer 0 when hp must be made) */
    void* hp; int64_t ldp_h;               /* PLANES: P3 image of the input [n][fin]                        */
    const int32_t* h_rows; int64_t n_res_rows;   /* PLANES layer 0: hp is the RESIDENT image, read through this row map */
    int make_hp;                           /* PLANES: 1 = convert x (fp32) into hp first                    */
    float* ahn;                            /* SMALLK: aggregated input [n][fin]                             */
    float* t;                              /* PLANES: [n][2 fout]; SMALLK: z [n][fout]                      */
    float* stats;                          /* [2 n]                                                         */
    float* y;                              /* fp32 output [n][fout] (NULL when only the next image is needed) */
    void* yp; int64_t ldp_y;               /* P3 image of the output for the next PLANES layer (or NULL)    */
    float* dy;                             /* [n][fout] gradient w.r.t. the output; dz in place             */
    void* dzp; void* qp; int64_t ldp_o;    /* PLANES: images [n][fout]                                      */
    void* ws_ln; int64_t ws_ln_bytes;      /* gte_ln_relu_bwd workspace                                     */
    void* ws_dw; int64_t ws_dw_bytes;      /* split-K workspace of the layer's dW                           */
    int64_t ldf;                           /* floats per row of y / dy / z (t: 2 ldf): fout rounded up to 16; 0 = fout */
    void* ahnp; int64_t ldp_ahn;           /* AGGFIRST: P3 image of the aggregated input [n][fin]; CACHED: the RESIDENT image of it */
} gte_step_layer;
typedef struct gte_step_plan {
    int n_hidden;                          /* hidden layers (1 .. 7), followed by the output layer          */
    gte_step_layer layer[7];
    int64_t out_fin, n_classes;            /* output layer                                                  */
    const float* W_out; const float* b_out; float* gW_out; float* gb_out;
    const float* h_out; int64_t ld_h_out;  /* fp32 input of the output layer (= y of the last hidden layer) */
    float* logits; float* tn; float* q_out; float* dl; float* dh_out;
    void* ce_part; int64_t ce_part_bytes; void* ws_nar; int64_t ws_nar_bytes;
    const int32_t* indptr; const int32_t* indices; const float* w_in;         /* in-edge CSR                */
    const int32_t* rindptr; const int32_t* rindices; const float* w_out;      /* out-edge CSR, weights x 1 / in_degree(dst) */
    int64_t n_nodes;
    const void* labels; int labels_f32; const float* class_weights; float grad_scale; float* out3;
    const void* wimg_descs; int n_wimg_descs;                                /* gte_p3_desc[]: the weight images */
    float* param; float* grad; float* exp_avg; float* exp_avg_sq; int64_t n_param;   /* fused Adam (NULL param: plain flush) */
    float* hyper; int64_t* step_counter; uint32_t* ticket;
    void* tail_ws; int64_t tail_ws_bytes;
    int fuse_ln_dx;                        /* bit 0: dX of a PLANES layer above a PLANES layer runs gte_gemm_p3_nt_ln_bwd;
                                              bit 1: the output layer's backward runs gte_sage_narrow_bwd_ln_p3;
                                              bit 2: the GEMM output layer (out_gemm) runs gte_head_agg_ce + gte_head_dlq_finish;
                                              bit 3: dX of layer 1 above a SMALLK layer 0 runs gte_gemm_p3_nt_smallk_bwd;
                                              bit 4: an AGGFIRST / CACHED input layer with fout <= 256 runs its LayerNorm forward as
                                                     the epilogue of its GEMM (gte_gemm_p3_nt[_rows2]_ln_fwd)                  */
    int wimg_fresh;                        /* the weight images already hold the current parameters: the forward skips their
                                              conversion launch (set by the caller after a step that returned *adam_fused & 2) */
    int wimg_in_fold;                      /* the fold + Adam launch also writes the weight images of the UPDATED parameters
                                              (gte_fold_defer_flush_adam_images); *adam_fused & 2 reports that it did        */
    /* output layer on the planes GEMMs (out_gemm = 1): logits / tn are columns 0.. / 16.. of ONE [n][32] fp32 buffer (logits =
     * its base, tn = logits + 16), dl / q_out likewise of a second one whose other columns stay zero; ld_lg = 32 */
    int out_gemm; int64_t ld_lg;           /* ld_lg: floats per row of logits / tn / dl / q_out (0 = n_classes)             */
    void* hp_out; int64_t ldp_hout;        /* P3 image of the output layer's input [n][out_fin] (yp of the last hidden layer) */
    void* wimg_out_fwd; int64_t ldp_wout_fwd;   /* P3 [32][out_fin]: rows 0.. = W_s rows, rows 16.. = W_n rows              */
    void* wimg_out_bwd; int64_t ldp_wout_bwd;   /* P3 [out_fin][32]: columns 0.. = W_s^T, 16.. = W_n^T                      */
    void* dlqp; int64_t ldp_dlq;           /* P3 image [n + 1][32] of the dl / q buffer                                     */
    void* ws_out; int64_t ws_out_bytes;    /* split-K workspace of dW_out: gte_gemm_p3_tn_workspace_bytes(C, 2 out_fin, out_fin, n) */
    void* ws_ce; int64_t ws_ce_bytes;      /* gte_weighted_ce workspace                                                     */
    void* ws_cs; int64_t ws_cs_bytes;      /* gte_colsum workspace                                                          */
    /* measurement (NULL: off): hipEvent_t handles, recorded on `stream` in front of ([2 i]) and behind ([2 i + 1]) the forward
     * transform GEMM of hidden layer i -- the roofline figure of a shape is timed inside the loop it belongs to */
    void* const* fwd_events;
} gte_step_plan;
/* *adam_fused: bit 0 = the optimiser step ran inside the fold launch, bit 1 = ... and it wrote the weight images */
int gte_gcnsage_step(const gte_step_plan* plan, int phase, int* adam_fused, void* stream);
/* The forward pass alone from the same plan (replaces `logits = model(g)` under no_grad: src/models/model_predict.py:141-147 and
 * the validation forward src/models/model_train.py:349-353): weight images, hidden layers, the output layer; logits [n, C] in
 * plan->logits.  The same kernels in the same order as the forward half of gte_gcnsage_step (bit-identical activations);
 * labels, gradients, optimiser fields and backward workspaces of the plan are ignored. */
int gte_gcnsage_forward(const gte_step_plan* plan, void* stream);

/* ---- deferred folds -------------------------------------------------------------------------------------------
 * Several entry points end with a small "sum the per-block partials" kernel (gte_ln_relu_bwd: column sums;
 * gte_sage_narrow_bwd: dW / dbias; split-K GEMMs behind gte_sage_linear_dw / gte_sage_qform_dw).  gte_gemm_f32 NEVER defers:
 * its result (e.g. dh of a backward) is read by the next kernel of the stream.  Between
 * gte_fold_defer_begin(stream) and gte_fold_defer_flush() (same host thread) those folds are queued instead of launched
 * and the flush runs all of them in ONE launch on `stream`.  While a deferral is open every such call needs its OWN
 * workspace (it holds the partials until the flush) and its results are not final before the flush.  Fixed summation
 * order: deterministic.  No counterpart in the reference (torch autograd launches one reduction per parameter). */
int gte_fold_defer_begin(void* stream);
int gte_fold_defer_flush(void);
/* Close the deferral and, when the queued folds write every element of the flat gradient [grad, grad + n) exactly once,
 * run them AND the optimiser step of gte_adam_step_dev (same state / step_counter / ticket contract, same arithmetic) in
 * ONE launch: the thread that folds a gradient element applies its Adam update.  *fused = 1 then.  Otherwise (a gradient
 * some producer wrote directly, a deferral that overflowed) the folds run as gte_fold_defer_flush would, *fused = 0, and the
 * caller launches gte_adam_step_dev itself.  grad still receives the folded gradient either way. */
int gte_fold_defer_flush_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                              int64_t* step_counter, unsigned* ticket, int* fused);
/* ... and, in the same launch, the P3 images of up to 12 sub-matrices of the UPDATED parameters (every `src` inside
 * [param, param + n); the descriptors of gte_p3_from_f32_batch): the thread that updates a parameter element writes its three
 * bf16 pieces into each image holding it -- bit for bit what gte_p3_from_f32_batch makes of the updated parameters, without
 * the launch in front of the next step's first GEMM.  Padding columns of the images are not touched (zero them once).
 * *fused = 3 when the step and the images were written, 1 when more than 12 images were asked for (or one with a single source column) (step applied, images not),
 * 0 as gte_fold_defer_flush_adam. */
int gte_fold_defer_flush_adam_images(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                     int64_t* step_counter, unsigned* ticket, const gte_p3_desc* images, int n_images, int* fused);

/* ---- transform-then-aggregate ("q-form") of a GcnSAGELayer --------------------------------------------------
 * replaces (reference src/components/graphs/models.py:53-72, `torch.cat((h, ah * norm), 1)` -> nn.Linear) where the
 * layer narrows: by linearity  z = h W_s^T + b + norm * A_w (h W_n^T), so the aggregation moves n_out columns
 * instead of n_feat.  Backward with q = A_w^T (norm * dz):  dW = [dz^T h | q^T h],  dh = dz W_s + q W_n.
 *   gte_sage_transform_fwd : t[:, 0:n_out] = x W_s^T + bias,  t[:, n_out:2 n_out] = x W_n^T      (one GEMM launch)
 *   gte_sage_qform_dw      : dW[n_out, 2 n_feat] = [dz^T x | q^T x]                               (split over nodes)
 *   gte_sage_qform_dx      : dx[nodes, n_feat]   = dz W_s + q W_n                                 (one GEMM, K = 2 n_out)
 * W is the layer weight [n_out][2 n_feat] (ldw >= 2 n_feat).  All matrices fp32 row-major with leading dimensions. */
int gte_sage_transform_fwd(const float* x, int64_t ldx, int64_t n_feat, const float* W, int64_t ldw, const float* bias,
                           int64_t n_out, float* t, int64_t ldt, int64_t n_nodes, void* stream);
int64_t gte_sage_qform_dw_workspace_bytes(int64_t n_out, int64_t n_feat, int64_t n_nodes);
int gte_sage_qform_dw(const float* dz, int64_t lddz, const float* q, int64_t ldq, const float* x, int64_t ldx,
                      int64_t n_feat, float* dW, int64_t lddw, int64_t n_out, int64_t n_nodes, void* workspace,
                      int64_t workspace_bytes, void* stream);
int gte_sage_qform_dx(const float* dz, int64_t lddz, const float* q, int64_t ldq, const float* W, int64_t ldw,
                      int64_t n_feat, int64_t n_out, float* dx, int64_t lddx, int64_t n_nodes, void* stream);

/* The class-count-wide output layer (hidden -> n_classes, no LayerNorm / activation: models.py:101-103) in
 * transform-then-aggregate form (n_out <= 16, n_feat <= 256; gte_sage_narrow_supported says so):
 *   fwd: t_self = h W[:, 0:F]^T + bias, t_neigh = h W[:, F:2F]^T;  the caller finishes with
 *        logits = t_self + mean-aggregate(t_neigh)  (gte_spmm_csr_accumulate, REDUCE_MEAN) -- by linearity equal to
 *        [h | norm * A_w h] W^T + b of models.py:53-72, with the aggregation on n_out columns instead of n_feat.
 *   bwd: given dl = dlogits and q = A_w^T(norm * dl) (gte_spmm_csr over the out-edge CSR), writes
 *        dh = dl W_s + q W_n (nullable), dW = [dl^T h | q^T h], dbias = colsum(dl).  Reads h once. */
int gte_sage_narrow_supported(int64_t n_feat, int64_t n_out);
int gte_sage_narrow_fwd(const float* h, int64_t ldh, int64_t n_feat, const float* W, int64_t ldw, const float* bias,
                        int64_t n_out, float* t_self, int64_t ld_self, float* t_neigh, int64_t ld_neigh,
                        int64_t n_nodes, void* stream);
/* gte_sage_narrow_fwd on the output of a LayerNorm(+ReLU) that has not been applied yet: z is the pre-LayerNorm output of
 * the layer below (models.py:63); the kernel writes y = relu?(LN(z)) (models.py:64-66) and stats = {mean[n], rstd[n]} for the
 * backward and multiplies the normalised rows in the same pass -- one launch and one pass over [n, n_feat] less than
 * gte_ln_relu_fwd + gte_sage_narrow_fwd.  n_feat % 8 == 0 (gte_sage_narrow_fwd_ln_supported). */
int gte_sage_narrow_fwd_ln_supported(int64_t n_feat, int64_t n_out);
int gte_sage_narrow_fwd_ln(const float* z, int64_t ldz, int64_t n_feat, const float* gamma, const float* beta, float eps,
                           int relu, float* y, int64_t ldy, float* stats, const float* W, int64_t ldw, const float* bias,
                           int64_t n_out, float* t_self, int64_t ld_self, float* t_neigh, int64_t ld_neigh, int64_t n_nodes,
                           void* stream);
int64_t gte_sage_narrow_bwd_workspace_bytes(int64_t n_nodes, int64_t n_feat, int64_t n_out);
int gte_sage_narrow_bwd(const float* dl, int64_t lddl, const float* q, int64_t ldq, const float* h, int64_t ldh,
                        int64_t n_feat, const float* W, int64_t ldw, int64_t n_out, float* dh, int64_t lddh,
                        float* dW, int64_t lddw, float* dbias, int64_t n_nodes,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* The fused head of the train step (class-count-wide output layer + weighted cross-entropy, model_train.py:171,327-328
 * after models.py:101-103).  gte_head_supported(n_feat, n_classes): n_classes <= 16, n_feat <= 256, n_feat % 8 == 0.
 *   gte_head_agg_ce : logits[v,:] (in: t_self + bias from gte_sage_narrow_fwd) += scale_v * sum_e w[e] t_neigh[src,:]  (the
 *        same sums, bit for bit, as gte_spmm_csr_accumulate), then per node the weighted CE terms and the UNNORMALISED
 *        gradient dl'[v,:] = w_y (softmax - onehot); per-block partials {sum w nll, sum w, #correct} go to ce_partial
 *        (>= gte_head_agg_ce_workspace_bytes(n_nodes), must stay untouched until gte_sage_narrow_bwd_ce has run).
 *   caller: q' = gte_spmm_csr(out-edge CSR, mean weights, dl')   (linear: the missing 1 / sum(w) commutes with it)
 *   gte_sage_narrow_bwd_ce : gte_sage_narrow_bwd on alpha * dl', alpha * q' with alpha = grad_scale / sum(w) folded from
 *        ce_partial inside the kernel; also writes out3 = {loss, sum w, #correct} like gte_weighted_ce.
 * Three launches (aggregation, CE partial, CE gradient) become one and the loss never needs a launch of its own. */
int gte_head_supported(int64_t n_feat, int64_t n_classes);
int64_t gte_head_agg_ce_workspace_bytes(int64_t n_nodes);
int gte_head_agg_ce(const int32_t* indptr, const int32_t* indices, const float* eweight, const float* t_neigh,
                    int64_t ld_neigh, float* logits, int64_t ld_logits, const void* labels, int labels_f32,
                    const float* class_weight, int64_t n_nodes, int64_t n_classes, int reduce, float* dl_unscaled,
                    int64_t lddl, void* ce_partial, int64_t ce_partial_bytes, void* stream);
int gte_sage_narrow_bwd_ce(const float* dl_unscaled, int64_t lddl, const float* q_unscaled, int64_t ldq, const float* h,
                           int64_t ldh, int64_t n_feat, const float* W, int64_t ldw, int64_t n_out, float* dh,
                           int64_t lddh, float* dW, int64_t lddw, float* dbias, int64_t n_nodes, void* workspace,
                           int64_t workspace_bytes, const void* ce_partial, float grad_scale, float* out3, void* stream);

/* gte_sage_narrow_bwd_ce (ce_partial may be NULL: then dl / q are final and out3 is untouched) fused with the
 * LayerNorm(+ReLU) backward of the layer below, whose output h = relu?(LN(z_below)) fed the output layer: instead of dh the
 * call writes dz_below = LN'(z_below)(mask . dh) and the parameter gradients dgamma / dbeta / dbias of the layer below
 * (each nullable) -- the [N, F] matrix dh is neither written nor read back by a separate gte_ln_relu_bwd.
 * ln_workspace >= gte_sage_narrow_bwd_ln_workspace_bytes(n_nodes, n_feat); its folds join an open deferral
 * (gte_fold_defer_begin) or run as one launch.  Same support rule as gte_head_supported. */
int64_t gte_sage_narrow_bwd_ln_workspace_bytes(int64_t n_nodes, int64_t n_feat);
/* Row form: the dh tile of every 32-row block goes through LDS and whole rows get the LayerNorm(+ReLU) backward with 16-byte
 * accesses and the arithmetic of gte_ln_relu_bwd; dz_below as fp32 AND as a P3 image (dzp3 nullable; n_feat % 16 == 0).  Bit for
 * bit gte_sage_narrow_bwd[_ce] + gte_ln_relu_bwd_p3.  (Round 2's accumulator-layout form gte_sage_narrow_bwd_ln and round 3's
 * option of forming q inside the kernel measured slower than what they replaced and were removed in round 4.) */
int gte_sage_narrow_bwd_ln_p3(const float* dl, int64_t lddl, const float* q, int64_t ldq, const float* h, int64_t ldh,
                              int64_t n_feat, const float* W, int64_t ldw, int64_t n_out, float* dz_below, int64_t lddz,
                              void* dzp3, int64_t ldp3, float* dW, int64_t lddw, float* dbias, int64_t n_nodes, void* workspace,
                              int64_t workspace_bytes, const void* ce_partial, float grad_scale, float* out3,
                              const float* z_below, int64_t ldz, const float* stats_below, const float* gamma_below,
                              const float* beta_below, int relu_below, float* dgamma_below, float* dbeta_below,
                              float* dbias_below, void* ln_workspace, int64_t ln_workspace_bytes, void* stream);
/* The output layer on PADDED hidden rows (round 5: hidden widths that are not a multiple of 8 -- int(calculate_hidden) = 100 / 139 /
 * 149 / 157 / 206 / 218 of the reference's scaled runs -- keep the narrow kernels instead of three planes GEMMs): h, z_below and
 * dz_below rows hold n_feat true columns and zeros up to n_pad (a multiple of 16, <= 256); W / dW are the reference's [C][2 n_feat].
 * The products run over n_pad columns against a weight image whose padding is zero (the same sums); the fused LayerNorm backward
 * runs over the n_feat true columns with the arithmetic of gte_ln_relu_bwd_p3 at that width and writes zeros into the padding of
 * dz_below and of its image (n_pad / 16 blocks).  Workspace sizes: the functions above with n_feat = n_pad. */
int gte_sage_narrow_pad_supported(int64_t n_feat, int64_t n_pad, int64_t n_out);
int gte_sage_narrow_fwd_pad(const float* h, int64_t ldh, int64_t n_feat, int64_t n_pad, const float* W, int64_t ldw, const float* bias,
                            int64_t n_out, float* t_self, int64_t ld_self, float* t_neigh, int64_t ld_neigh, int64_t n_nodes,
                            void* stream);
int gte_sage_narrow_bwd_ln_p3_pad(const float* dl, int64_t lddl, const float* q, int64_t ldq, const float* h, int64_t ldh,
                                  int64_t n_feat, int64_t n_pad, const float* W, int64_t ldw, int64_t n_out, float* dz_below,
                                  int64_t lddz, void* dzp3, int64_t ldp3, float* dW, int64_t lddw, float* dbias, int64_t n_nodes,
                                  void* workspace, int64_t workspace_bytes, const void* ce_partial, float grad_scale, float* out3,
                                  const float* z_below, int64_t ldz, const float* stats_below, const float* gamma_below,
                                  const float* beta_below, int relu_below, float* dgamma_below, float* dbeta_below,
                                  float* dbias_below, void* ln_workspace, int64_t ln_workspace_bytes, void* stream);

/* LayerNorm + ReLU alone (row-wise over n_out):  y = relu?(gamma * (z - mean) * rstd + beta).
 * replaces models.py:64-66 when the caller ran the linear part separately.  In place (y == z) is
 * allowed.  stats (nullable): f32[2*M] = mean, rstd. */
int gte_ln_relu_fwd(const float* z, int64_t ldz, const float* gamma, const float* beta, float eps, int relu,
                    float* y, int64_t ldy, float* stats, int64_t M, int64_t n_out, void* stream);

/* ... on PADDED rows of any width up to 1024 (ldz, ldy >= n_out rounded up to 4 floats), y as fp32 (nullable) and / or as a
 * P3 image (nullable; its columns up to the next multiple of 16 are written as zeros): the LayerNorm(+ReLU) of an
 * aggregate-first planes layer (GTE_LAYER_AGGFIRST).  Same two-pass statistics over the n_out true columns. */
int gte_ln_relu_fwd_p3(const float* z, int64_t ldz, const float* gamma, const float* beta, float eps, int relu, float* y,
                       int64_t ldy, void* yp3, int64_t ldyp3, float* stats, int64_t M, int64_t n_out, void* stream);

/* Backward of LayerNorm+ReLU (+ bias grad):  given dy, the saved z and stats, writes
 *   dz[M,n_out], and WRITES the column sums dgamma, dbeta, dbias (f32[n_out], nullable; no zero-init needed).
 * With gamma == NULL it is the backward of (relu?)(z) only.  dz may alias dy unless gamma != NULL and
 * n_out > 1024 (wide rows are re-read).
 * workspace: gte_ln_relu_bwd_workspace_bytes(M, n_out). */
int64_t gte_ln_relu_bwd_workspace_bytes(int64_t M, int64_t n_out);
int gte_ln_relu_bwd(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* stats,
                    const float* gamma, const float* beta, int relu,
                    float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dbias,
                    int64_t M, int64_t n_out, void* workspace, int64_t workspace_bytes, void* stream);

/* General fp32 MFMA GEMM used by the backward:  C[M,N] (+)= op(A)[M,K] * op(B)[K,N]
 *   trans_a = 0: A is [M,K] row-major (lda);  1: A is stored [K,M] (lda) and used transposed
 *   trans_b = 0: B is [K,N] row-major (ldb);  1: B is stored [N,K] (ldb) and used transposed
 * accumulate != 0 adds into C.  K may be huge (dW = dZ^T X reduces over the nodes): the kernel
 * splits K across workgroups and reduces the partial slabs deterministically in `workspace`
 * (gte_gemm_workspace_bytes).  replaces autograd of nn.Linear (models.py:63). */
int64_t gte_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gte_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                 const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, int accumulate,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A10  loss and optimiser
 * gte_weighted_ce: replaces nn.CrossEntropyLoss(weight)(logits, labels.long()) forward+backward
 *   (model_train.py:171,327) and the accuracy count (:328).
 *   loss = sum_i w[y_i] * nll_i / sum_i w[y_i]   (w == NULL => plain mean)
 *   dlogits (nullable) = grad_scale * d loss / d logits.
 *   out3: f32[3] = {loss, sum of weights, #correct (argmax == label)}.
 *   labels: int64[n]  (labels_f32 != 0: float32 holding integers, loader.py:350-354).
 * workspace: gte_weighted_ce_workspace_bytes(n).
 * ---------------------------------------------------------------------------------------- */
int64_t gte_weighted_ce_workspace_bytes(int64_t n_nodes);
int gte_weighted_ce(const float* logits, int64_t ld, const void* labels, int labels_f32,
                    const float* class_weight, int64_t n_nodes, int n_classes, float grad_scale,
                    float* dlogits, int64_t lddl, float* out3,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* out[c] = sum_r x[r][c], c < n_cols <= 64: the bias gradient of the output layer (colsum of dlogits; autograd of the
 * nn.Linear bias, models.py:101-103) when its backward runs on the planes GEMMs.  Block partials in a fixed order; the fold
 * joins an open fold deferral.  workspace: gte_colsum_workspace_bytes(n_rows, n_cols). */
int64_t gte_colsum_workspace_bytes(int64_t n_rows, int64_t n_cols);
int gte_colsum(const float* x, int64_t ldx, int64_t n_rows, int64_t n_cols, float* out, void* workspace,
               int64_t workspace_bytes, void* stream);

/* gte_adam_step: replaces torch.optim.Adam(lr, weight_decay).step() (model_train.py:168,332) on one
 * flat fp32 buffer: g' = grad_scale*g + weight_decay*p (L2-coupled, NOT AdamW);
 * m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 * `step` is t (1-based).  lr is read from the host value. */
int gte_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                  float grad_scale, void* stream);
/* The same update with everything that changes from step to step read from DEVICE memory, so the launch can be captured
 * in a HIP graph.  state (8 floats, read AND advanced by the call) = {lr, beta1, beta2, eps, weight_decay, grad_scale,
 * bc1 = 1 - beta1^t, sqrt(bc2) = sqrt(1 - beta2^t)} for t = *step_counter + 1; the last block to finish sets
 * *step_counter = t and the two bias corrections for t + 1 (double arithmetic, like the host-scalar version).
 * `ticket` is gte_adam_ticket_bytes() bytes of zero-initialised scratch that the call returns to zero (sharded completion
 * counters: one counter for a thousand workgroups is a 12 - 20 us queue).  The caller initialises state and counter
 * consistently (and may rewrite lr at any time between launches). */
int64_t gte_adam_ticket_bytes(void);
int gte_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                      int64_t* step_counter, unsigned* ticket, void* stream);
/* gte_adam_step_dev + the P3 images of the UPDATED parameters in the same launch (images: sub-matrices of `param`, as
 * gte_fold_defer_flush_adam_images takes them): the optimiser launch of the data-parallel step (behind the all-reduce of
 * model_train.py:327-332's gradients, where the fold launch cannot apply the update) leaves the weight images of the next forward
 * behind.  *wrote = 1 when the images were written (a list of more than 12 images is skipped). */
int gte_adam_step_dev_images(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                             int64_t* step_counter, uint32_t* ticket, const gte_p3_desc* images, int n_images, int* wrote,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * A13  multi-head graph attention (BASELINE.json configs[2]).  The reference has NO GAT: the
 * layer is the standard formulation (DGL GATConv semantics), oracle = oracle/gat_cpu.py, parity unpinned.
 *   z [N, H*D] (= X W, computed with gte_gemm_f32);  heads <= 8, heads*dim <= 1024
 * gte_gat_scores: el[v,h] = <a_l[h], z[v,h]>, er likewise; optionally writes a bf16 copy of z for the gathers.
 * gte_gat_aggregate_fwd: e = LeakyReLU_0.2(el[u] + er[v]) over in-edges, online softmax, out[v] = sum alpha z[u] (+bias);
 *   saves the softmax max / sum per (node, head) for the backward.  dtype = storage of the gathered z (f32 / bf16).
 * gte_gat_aggregate_bwd: given dout, writes dz (incl. the el/er paths), d a_l, d a_r, d bias; ds [E,H], der, del [N,H]
 *   are caller-provided scratch outputs; pos_in[i] = position in the in-edge CSR of the i-th out-edge-CSR entry.
 * ---------------------------------------------------------------------------------------- */
int gte_gat_scores(const float* z, int64_t ldz, const float* a_l, const float* a_r, float* el, float* er,
                   void* z_bf16 /* nullable */, int64_t ldzb, int64_t n_nodes, int heads, int dim, void* stream);
int gte_gat_aggregate_fwd(const int32_t* indptr, const int32_t* indices, const void* z, int64_t ldz, int dtype,
                          const float* el, const float* er, const float* bias /* nullable */, float* out, int64_t ldo,
                          float* smax, float* ssum, int64_t n_nodes, int heads, int dim, void* stream);
/* The same with an epilogue (BASELINE configs[2]): activation 1 = ELU applied to (aggregate + bias) -- the activation between
 * GAT layers; out_bf16 (nullable): bf16 copy of that output, the next layer's gte_gemm_bf16_nt operand; out_mean (nullable):
 * [n, dim] mean over the heads + mean_bias[dim] -- the output layer (then `out` may be NULL).
 * gte_gat_dout_prepare turns the gradient w.r.t. such an output back into the gradient gte_gat_aggregate_bwd takes:
 * dfull[v, f] = (mean_heads ? dout[v, f % dim] / heads : dout[v, f]) * (act_out ? ELU'(act_out[v, f]) : 1). */
int gte_gat_aggregate_fwd_ex(const int32_t* indptr, const int32_t* indices, const void* z, int64_t ldz, int dtype,
                             const float* el, const float* er, const float* bias, float* out, int64_t ldo, float* smax,
                             float* ssum, int64_t n_nodes, int heads, int dim, int activation, void* out_bf16, int64_t ldob,
                             float* out_mean, int64_t ldom, const float* mean_bias, void* stream);
int gte_gat_dout_prepare(const float* dout, int64_t lddo, const float* act_out, int64_t ldao, float* dfull, int64_t lddf,
                         int64_t n_nodes, int heads, int dim, int mean_heads, void* stream);
int64_t gte_gat_bwd_workspace_bytes(int64_t n_nodes, int heads, int dim);
int gte_gat_aggregate_bwd(const int32_t* indptr, const int32_t* indices, const int32_t* rindptr, const int32_t* rindices,
                          const int32_t* pos_in, const void* z, int64_t ldz, int dtype, const float* z_f32, int64_t ldzf,
                          const float* el, const float* er, const float* smax, const float* ssum,
                          const float* a_l, const float* a_r, const float* dout, int64_t lddo,
                          float* ds, float* der, float* del, float* dz, int64_t lddz,
                          float* da_l, float* da_r, float* dbias, int64_t n_nodes, int heads, int dim,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* The same with gte_gat_dout_prepare folded in: dfull (nullable; [n, heads * dim] scratch) receives the effective gradient
 * (mean_heads ? dout[v, f % dim] / heads : dout[v, f]) * (act_out ? ELU'(act_out[v, f]) : 1), formed by the destination-side
 * kernel itself where the row-layout kernels apply (heads <= 4, dim <= 64: no separate pass over the three [n, heads * dim]
 * arrays) and by a gte_gat_dout_prepare launch otherwise.  dfull == NULL: dout is used as given (gte_gat_aggregate_bwd). */
int gte_gat_aggregate_bwd_ex(const int32_t* indptr, const int32_t* indices, const int32_t* rindptr, const int32_t* rindices,
                             const int32_t* pos_in, const void* z, int64_t ldz, int dtype, const float* z_f32, int64_t ldzf,
                             const float* el, const float* er, const float* smax, const float* ssum,
                             const float* a_l, const float* a_r, const float* dout, int64_t lddo,
                             const float* act_out, int64_t ldao, int mean_heads, float* dfull, int64_t lddf,
                             float* ds, float* der, float* del, float* dz, int64_t lddz,
                             float* da_l, float* da_r, float* dbias, int64_t n_nodes, int heads, int dim,
                             void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GTE_H_ */
