// One optimisation step of GcnSAGE from a prepared plan: gte_gcnsage_step (include/gte.h).
//
// replaces the batch loop body of the reference (src/models/model_train.py:320-332) as ONE host call.  Host code only: the
// arithmetic is the library's own entry points, issued in the order models/engine.py issues them (the two paths are
// interchangeable and bit-identical: tests/test_gpu_step_plan.py).  What this buys is host time: a step is ~17 launches; from
// Python through ctypes each costs ~10 us of interpreter + marshalling, which bounded the BBOX-only configurations (F0 = 13:
// ~0.25 ms of host time per 0.37 ms step).
#include "gte_common.h"

#include <stdlib.h>

namespace {

#define GTE_TRY(call)            \
    do {                         \
        const int rc_ = (call);  \
        if (rc_ != GTE_OK) return rc_; \
    } while (0)

// leading dimension (floats) of a layer's fp32 row buffers: fout rounded up to 16 (0 in the plan = fout itself)
inline int64_t ldf(const gte_step_layer& L) { return L.ldf > 0 ? L.ldf : L.fout; }
inline int64_t ld_lg(const gte_step_plan& p) { return p.ld_lg > 0 ? p.ld_lg : p.n_classes; }

// the GEMM output layer with the fused head: aggregation + loss in one launch (dl without 1 / sum w), the scaling, the [dl | q]
// image, the bias gradient and out3 in another (gte_head_agg_ce + gte_head_dlq_finish)
bool head_gemm_fused(const gte_step_plan& p) {
    return p.out_gemm && (p.fuse_ln_dx & 4) && p.n_classes <= 16 && ld_lg(p) >= 32 && p.n_nodes > 0 &&
           p.ws_ce_bytes >= gte_head_agg_ce_workspace_bytes(p.n_nodes) && p.ws_cs_bytes >= gte_head_dlq_finish_workspace_bytes(p.n_nodes);
}


// the narrow output kernels on a hidden width that is not a multiple of 8 (padded rows of ld_h_out floats: the general plan)
bool narrow_padded(const gte_step_plan& p) {
    return !p.out_gemm && p.out_fin % 8 != 0 && p.ld_h_out > p.out_fin && gte_sage_narrow_pad_supported(p.out_fin, p.ld_h_out, p.n_classes);
}

// weight images + the hidden layers (the output layer's input is p.h_out / p.hp_out afterwards).  fwd_only: nothing is kept for a
// backward -- a planes layer whose output is consumed as an image (by the next planes layer, or by the output layer's GEMM) does
// not write its fp32 rows (a sixth of the aggregation + LayerNorm kernel's bytes)
int forward_hidden(const gte_step_plan& p, void* st, bool fwd_only = false) {
    const int64_t n = p.n_nodes;
    if (p.n_wimg_descs > 0 && !p.wimg_fresh) {
        const gte_p3_desc* d = reinterpret_cast<const gte_p3_desc*>(p.wimg_descs);
        for (int k = 0; k < p.n_wimg_descs; k += 16)
            GTE_TRY(gte_p3_from_f32_batch(d + k, p.n_wimg_descs - k < 16 ? p.n_wimg_descs - k : 16, st));
    }
    auto mark = [&](int k) {        // (measurement: an event in front of / behind a layer's forward GEMM)
        if (p.fwd_events && p.fwd_events[k]) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(p.fwd_events[k]), gte::as_stream(st));
    };
    // y of hidden layer i as fp32 rows is read by nobody when the layer above is a PLANES layer on this layer's image (its forward
    // and its dW take the image; every LayerNorm backward recomputes the ReLU mask from z); the last hidden layer's rows feed the
    // narrow output kernels (a forward-only pass with the GEMM output layer reads the image there too)
    auto y_unread = [&](int i) {
        if (i + 1 < p.n_hidden) return p.layer[i + 1].kind == GTE_LAYER_PLANES && !p.layer[i + 1].make_hp;
        return fwd_only && p.out_gemm != 0;
    };
    for (int i = 0; i < p.n_hidden; ++i) {
        const gte_step_layer& L = p.layer[i];
        const int64_t ld = ldf(L);
        if (L.kind == GTE_LAYER_SMALLK) {
            GTE_TRY(gte_spmm_csr(p.indptr, p.indices, p.w_in, L.x, L.ldx, L.ahn, L.fin, n, L.fin, GTE_F32, GTE_REDUCE_MEAN, st));
            // (the one-pass backward recomputes z from the 2 fin inputs per row: nothing to save)
            float* zs = gte_sage_smallk_bwd_supported(2 * L.fin, L.fout) ? nullptr : L.t;
            if (L.yp)          // the next layer's input image straight from this kernel (y itself is not needed then)
                GTE_TRY(gte_sage_linear_fwd_p3(L.x, L.ldx, L.fin, L.ahn, L.fin, L.fin, L.W, 2 * L.fin, L.bias, L.gamma, L.beta, L.eps, L.relu,
                                               zs, L.fout, L.stats, L.y, L.fout, L.yp, L.ldp_y, n, L.fout, st));
            else
                GTE_TRY(gte_sage_linear_fwd(L.x, L.ldx, L.fin, L.ahn, L.fin, L.fin, L.W, 2 * L.fin, L.bias, L.gamma, L.beta, L.eps, L.relu, zs,
                                            L.fout, L.stats, L.y, L.fout, n, L.fout, st));
            continue;
        }
        if (L.kind == GTE_LAYER_AGGFIRST) {
            // z = [x | mean-aggregate(x)] W^T + b on the planes GEMM (two K segments), then LayerNorm + ReLU: the input layer of a
            // model whose hidden width exceeds its input width (aggregating fin columns is the cheaper order, models.py:53-72)
            GTE_TRY(gte_p3_from_f32(L.x, L.ldx, n, L.fin, 0, L.hp, L.ldp_h, st));
            GTE_TRY(gte_spmm_csr_p3(p.indptr, p.indices, p.w_in, L.x, L.ldx, L.ahnp, L.ldp_ahn, n, L.fin, GTE_REDUCE_MEAN, st));
            mark(2 * i);
            if ((p.fuse_ln_dx & 16) && gte_gemm_p3_nt_ln_fwd_supported(L.fout)) {      // LayerNorm + ReLU as the GEMM's epilogue
                GTE_TRY(gte_gemm_p3_nt_ln_fwd(L.hp, L.ldp_h, L.fin, L.ahnp, L.ldp_ahn, L.fin, L.wimg_fwd, L.ldp_wfwd, L.bias, L.gamma, L.beta,
                                              L.eps, L.relu, L.t, ld, L.y, ld, L.yp, L.ldp_y, L.stats, n, L.fout, st));
                mark(2 * i + 1);
                continue;
            }
            GTE_TRY(gte_gemm_p3_nt(L.hp, L.ldp_h, L.fin, L.ahnp, L.ldp_ahn, L.fin, L.wimg_fwd, L.ldp_wfwd, L.bias, L.fout, L.t, ld, n, L.fout,
                                   0, 0, st));
            mark(2 * i + 1);
            GTE_TRY(gte_ln_relu_fwd_p3(L.t, ld, L.gamma, L.beta, L.eps, L.relu, L.y, ld, L.yp, L.ldp_y, L.stats, n, L.fout, st));
            continue;
        }
        if (L.kind == GTE_LAYER_CACHED) {
            // z = [x | ahn] W^T + b with x AND its mean aggregate read from their RESIDENT images through the batch's row map (the
            // aggregate of the input is page-local and constant: cached when the pages were loaded), then LayerNorm + ReLU
            mark(2 * i);
            float* const y3 = (L.yp && (fwd_only || y_unread(i))) ? nullptr : L.y;
            // (h_rows == NULL: the two images hold the graph's own rows -- a graph that is evaluated again and again, e.g. the
            // validation graph of train(), with its images made once)
            if ((p.fuse_ln_dx & 16) && gte_gemm_p3_nt_ln_fwd_supported(L.fout)) {      // the layer's whole forward in ONE launch
                if (L.h_rows)
                    GTE_TRY(gte_gemm_p3_nt_rows2_ln_fwd(L.hp, L.ldp_h, L.ahnp, L.ldp_ahn, L.fin, L.h_rows, L.n_res_rows, L.wimg_fwd,
                                                        L.ldp_wfwd, L.bias, L.gamma, L.beta, L.eps, L.relu, L.t, ld, y3, ld, L.yp, L.ldp_y,
                                                        L.stats, n, L.fout, st));
                else
                    GTE_TRY(gte_gemm_p3_nt_ln_fwd(L.hp, L.ldp_h, L.fin, L.ahnp, L.ldp_ahn, L.fin, L.wimg_fwd, L.ldp_wfwd, L.bias, L.gamma,
                                                  L.beta, L.eps, L.relu, L.t, ld, y3, ld, L.yp, L.ldp_y, L.stats, n, L.fout, st));
                mark(2 * i + 1);
                continue;
            }
            if (L.h_rows)
                GTE_TRY(gte_gemm_p3_nt_rows2(L.hp, L.ldp_h, L.ahnp, L.ldp_ahn, L.fin, L.h_rows, L.n_res_rows, L.wimg_fwd, L.ldp_wfwd, L.bias,
                                             L.fout, L.t, ld, n, L.fout, 0, 0, st));
            else
                GTE_TRY(gte_gemm_p3_nt(L.hp, L.ldp_h, L.fin, L.ahnp, L.ldp_ahn, L.fin, L.wimg_fwd, L.ldp_wfwd, L.bias, L.fout, L.t, ld, n,
                                       L.fout, 0, 0, st));
            mark(2 * i + 1);
            GTE_TRY(gte_ln_relu_fwd_p3(L.t, ld, L.gamma, L.beta, L.eps, L.relu, y3, ld, L.yp, L.ldp_y, L.stats, n, L.fout, st));
            continue;
        }
        if (L.make_hp) GTE_TRY(gte_p3_from_f32(L.x, L.ldx, n, L.fin, 0, L.hp, L.ldp_h, st));
        // t = [t_self | t_neigh], each half ld columns wide (the weight image holds zero rows behind the fout rows of a half)
        mark(2 * i);
        if (L.h_rows)
            GTE_TRY(gte_gemm_p3_nt_rows(L.hp, L.ldp_h, L.fin, L.h_rows, L.n_res_rows, L.wimg_fwd, L.ldp_wfwd, L.bias, L.fout, L.t, 2 * ld,
                                        n, 2 * ld, 0, 0, st));
        else
            GTE_TRY(gte_gemm_p3_nt(L.hp, L.ldp_h, L.fin, nullptr, 0, 0, L.wimg_fwd, L.ldp_wfwd, L.bias, L.fout, L.t, 2 * ld, n,
                                   2 * ld, 0, 0, st));
        mark(2 * i + 1);
        float* const y = (L.yp && y_unread(i)) ? nullptr : L.y;
        GTE_TRY(gte_spmm_csr_accumulate_ln_p3(p.indptr, p.indices, p.w_in, L.t + ld, 2 * ld, L.t, 2 * ld, n, L.fout,
                                              GTE_REDUCE_MEAN, L.gamma, L.beta, L.eps, L.relu, y, ld, L.yp, L.ldp_y, L.stats, st));
    }
    return GTE_OK;
}

// the output layer's two products: logits' = h W_s^T + b, t_neigh = h W_n^T
int forward_out_products(const gte_step_plan& p, void* st) {
    const int64_t n = p.n_nodes, C = p.n_classes, lg = ld_lg(p);
    if (p.out_gemm)            // hidden widths the narrow kernels do not cover: one planes GEMM with N = 32 (rows 0.. = W_s, 16.. = W_n)
        return gte_gemm_p3_nt(p.hp_out, p.ldp_hout, p.out_fin, nullptr, 0, 0, p.wimg_out_fwd, p.ldp_wout_fwd, p.b_out, C, p.logits, lg, n,
                              32, 0, 0, st);
    if (narrow_padded(p))      // hidden rows padded to 16 floats (zeros): the narrow kernels over the padded width
        return gte_sage_narrow_fwd_pad(p.h_out, p.ld_h_out, p.out_fin, p.ld_h_out, p.W_out, 2 * p.out_fin, p.b_out, C, p.logits, lg, p.tn,
                                       lg, n, st);
    return gte_sage_narrow_fwd(p.h_out, p.ld_h_out, p.out_fin, p.W_out, 2 * p.out_fin, p.b_out, C, p.logits, lg, p.tn, lg, n, st);
}

int forward(const gte_step_plan& p, void* st) {
    const int64_t n = p.n_nodes;
    GTE_TRY(forward_hidden(p, st));
    const int64_t C = p.n_classes, lg = ld_lg(p);
    GTE_TRY(forward_out_products(p, st));
    if (p.out_gemm) {
        if (head_gemm_fused(p))
            return gte_head_agg_ce(p.indptr, p.indices, p.w_in, p.tn, lg, p.logits, lg, p.labels, p.labels_f32, p.class_weights, n, C,
                                   GTE_REDUCE_MEAN, p.dl, lg, p.ws_ce, p.ws_ce_bytes, st);
        GTE_TRY(gte_spmm_csr_accumulate(p.indptr, p.indices, p.w_in, p.tn, lg, p.logits, lg, n, C, GTE_F32, GTE_REDUCE_MEAN, st));
        return gte_weighted_ce(p.logits, lg, p.labels, p.labels_f32, p.class_weights, n, (int)C, p.grad_scale, p.dl, lg, p.out3, p.ws_ce,
                               p.ws_ce_bytes, st);
    }
    GTE_TRY(gte_head_agg_ce(p.indptr, p.indices, p.w_in, p.tn, lg, p.logits, lg, p.labels, p.labels_f32, p.class_weights, n, C,
                            GTE_REDUCE_MEAN, p.dl, lg, p.ce_part, p.ce_part_bytes, st));
    return GTE_OK;
}

// layer 0 is a short-input layer whose whole backward rides in the epilogue of layer 1's dX (gte_gemm_p3_nt_smallk_bwd)
bool smallk_in_dx(const gte_step_plan& p) {
    if (!(p.fuse_ln_dx & 8) || p.n_hidden < 2) return false;
    const gte_step_layer& B = p.layer[0];
    const gte_step_layer& L = p.layer[1];
    return B.kind == GTE_LAYER_SMALLK && L.kind == GTE_LAYER_PLANES && gte_sage_smallk_bwd_supported(2 * B.fin, B.fout) &&
           gte_gemm_p3_nt_smallk_bwd_supported(2 * B.fin, L.fin) && ldf(B) == B.fout;
}

// z of a hidden layer (the operand of its LayerNorm backward) and its leading dimension
inline const float* z_of(const gte_step_layer& L) { return L.t; }
inline int64_t ldz_of(const gte_step_layer& L) { return L.kind == GTE_LAYER_PLANES ? 2 * ldf(L) : ldf(L); }

// backward of the output layer and of hidden layers n_hidden - 1 .. 1, and of layer 0 up to its weight-gradient GEMM
int backward_a(const gte_step_plan& p, void* st) {
    static const bool keep_dz0 = GTE_MEASURE_INT("GTE_STEP_KEEP_DZ0", 0) != 0;      // (round 5 before the skip)
    const int64_t n = p.n_nodes, C = p.n_classes, lg = ld_lg(p);
    bool ln_done = false, smallk_done = false;
    const gte_step_layer& T = p.layer[p.n_hidden - 1];
    if (p.out_gemm) {
        // q = A_w^T (norm dl); dl and q as ONE image [n][32] (dl in block 0, q in block 1); dW_out = [dl^T h | q^T h] (two column
        // segments of one TN launch, M = n_classes), dbias = colsum(dl), dh = [dl | q] [W_s^T | W_n^T] (K = 32)
        GTE_TRY(gte_spmm_csr(p.rindptr, p.rindices, p.w_out, p.dl, lg, p.q_out, lg, n, C, GTE_F32, GTE_REDUCE_SUM, st));
        if (head_gemm_fused(p)) {
            // (q from the aggregation launch above: gte_head_dlq_finish can form it itself from the out-edge CSR, one launch less,
            // but a thread per node walking its out-edges one after the other measured 14.7 us against 6 + 6 for the two launches)
            GTE_TRY(gte_head_dlq_finish(nullptr, nullptr, nullptr, p.dl, lg, n, C, p.ws_ce, p.grad_scale, p.out3, p.dlqp, p.ldp_dlq,
                                        p.gb_out, p.ws_cs, p.ws_cs_bytes, st));
        } else {
            GTE_TRY(gte_p3_from_f32(p.dl, lg, n, 32, 0, p.dlqp, p.ldp_dlq, st));
            GTE_TRY(gte_colsum(p.dl, lg, n, C, p.gb_out, p.ws_cs, p.ws_cs_bytes, st));
        }
        GTE_TRY(gte_gemm_p3_tn(p.dlqp, p.ldp_dlq, static_cast<const char*>(p.dlqp) + 96, p.ldp_dlq, p.hp_out, p.ldp_hout, nullptr, 0,
                               p.out_fin, p.gW_out, 2 * p.out_fin, C, 2 * p.out_fin, n, p.ws_out, p.ws_out_bytes, st));
        if ((p.fuse_ln_dx & 1) && T.kind != GTE_LAYER_SMALLK && gte_gemm_p3_nt_ln_bwd_supported(p.out_fin) && T.dy == p.dh_out &&
            (p.out_fin % 16 == 0 || ldf(T) >= ((p.out_fin + 3) & ~(int64_t)3))) {
            // ... with the LayerNorm(+ReLU) backward of the last hidden layer as its epilogue (hidden widths up to 256)
            GTE_TRY(gte_gemm_p3_nt_ln_bwd(p.dlqp, p.ldp_dlq, 32, nullptr, 0, 0, p.wimg_out_bwd, p.ldp_wout_bwd, z_of(T), ldz_of(T), T.stats,
                                          T.gamma, T.beta, T.relu, T.dy, ldf(T), T.dzp, T.ldp_o, T.ggamma, T.gbeta, T.gbias, n, p.out_fin,
                                          T.ws_ln, T.ws_ln_bytes, st));
            ln_done = true;
        } else {
            GTE_TRY(gte_gemm_p3_nt(p.dlqp, p.ldp_dlq, 32, nullptr, 0, 0, p.wimg_out_bwd, p.ldp_wout_bwd, nullptr, 0, p.dh_out, p.ld_h_out, n,
                                   p.out_fin, 0, 0, st));
        }
    } else if (narrow_padded(p)) {
        // ... the same on padded rows (hidden widths that are not a multiple of 8): LayerNorm backward of the last hidden layer over
        // its true width, partials written compact
        GTE_TRY(gte_spmm_csr(p.rindptr, p.rindices, p.w_out, p.dl, C, p.q_out, C, n, C, GTE_F32, GTE_REDUCE_SUM, st));
        GTE_TRY(gte_sage_narrow_bwd_ln_p3_pad(p.dl, C, p.q_out, C, p.h_out, p.ld_h_out, p.out_fin, p.ld_h_out, p.W_out, 2 * p.out_fin, C,
                                              p.dh_out, p.ld_h_out, T.dzp, T.ldp_o, p.gW_out, 2 * p.out_fin, p.gb_out, n, p.ws_nar,
                                              p.ws_nar_bytes, p.ce_part, p.grad_scale, p.out3, z_of(T), ldz_of(T), T.stats, T.gamma, T.beta,
                                              T.relu, T.ggamma, T.gbeta, T.gbias, T.ws_ln, T.ws_ln_bytes, st));
        ln_done = true;
    } else if ((p.fuse_ln_dx & 2) && T.kind == GTE_LAYER_PLANES && T.fout % 16 == 0 && gte_head_supported(p.out_fin, C)) {
        // the output layer's backward runs the LayerNorm(+ReLU) backward of the last hidden layer on the dh tile of every row block
        GTE_TRY(gte_spmm_csr(p.rindptr, p.rindices, p.w_out, p.dl, C, p.q_out, C, n, C, GTE_F32, GTE_REDUCE_SUM, st));
        GTE_TRY(gte_sage_narrow_bwd_ln_p3(p.dl, C, p.q_out, C, p.h_out, p.ld_h_out, p.out_fin, p.W_out, 2 * p.out_fin, C,
                                          p.dh_out, p.ld_h_out, T.dzp, T.ldp_o, p.gW_out, 2 * p.out_fin, p.gb_out, n, p.ws_nar, p.ws_nar_bytes,
                                          p.ce_part, p.grad_scale, p.out3, T.t, 2 * ldf(T), T.stats, T.gamma, T.beta, T.relu, T.ggamma,
                                          T.gbeta, T.gbias, T.ws_ln, T.ws_ln_bytes, st));
        ln_done = true;
    } else {
        GTE_TRY(gte_spmm_csr(p.rindptr, p.rindices, p.w_out, p.dl, C, p.q_out, C, n, C, GTE_F32, GTE_REDUCE_SUM, st));
        GTE_TRY(gte_sage_narrow_bwd_ce(p.dl, C, p.q_out, C, p.h_out, p.ld_h_out, p.out_fin, p.W_out, 2 * p.out_fin, C, p.dh_out, p.ld_h_out,
                                       p.gW_out, 2 * p.out_fin, p.gb_out, n, p.ws_nar, p.ws_nar_bytes, p.ce_part, p.grad_scale, p.out3, st));
    }
    for (int i = p.n_hidden - 1; i >= 0; --i) {
        const gte_step_layer& L = p.layer[i];
        const int64_t ld = ldf(L);
        if (L.kind == GTE_LAYER_SMALLK) {
            if (smallk_done) continue;
            if (!gte_sage_smallk_bwd_supported(2 * L.fin, L.fout))
                GTE_TRY(gte_ln_relu_bwd(L.dy, L.fout, L.t, L.fout, L.stats, L.gamma, L.beta, L.relu, L.dy, L.fout, L.ggamma, L.gbeta, L.gbias, n,
                                        L.fout, L.ws_ln, L.ws_ln_bytes, st));
            continue;                                  // (layer 0: its dW -- or its whole one-pass backward -- is phase 2)
        }
        if (!ln_done)          // (else: the dX launch of the layer above ran this layer's LayerNorm backward as its epilogue)
            GTE_TRY(gte_ln_relu_bwd_p3(L.dy, ld, z_of(L), ldz_of(L), L.stats, L.gamma, L.beta, L.relu, L.dy, ld, L.dzp, L.ldp_o, L.ggamma,
                                       L.gbeta, L.gbias, n, L.fout, L.ws_ln, L.ws_ln_bytes, st));
        ln_done = false;
        if (L.kind == GTE_LAYER_AGGFIRST || L.kind == GTE_LAYER_CACHED) break;       // (layer 0: dW = dz^T [x | ahn] is the step's last GEMM: phase 2)
        GTE_TRY(gte_spmm_csr_p3(p.rindptr, p.rindices, p.w_out, L.dy, ld, L.qp, L.ldp_o, n, L.fout, GTE_REDUCE_SUM, st));
        if (i == 0) break;                             // layer 0's dW is the step's last GEMM: phase 2
        GTE_TRY(gte_gemm_p3_tn(L.dzp, L.ldp_o, L.qp, L.ldp_o, L.hp, L.ldp_h, nullptr, 0, L.fin, L.gW, 2 * L.fin, L.fout, 2 * L.fin, n, L.ws_dw,
                               L.ws_dw_bytes, st));
        const gte_step_layer& B = p.layer[i - 1];
        if (i == 1 && smallk_in_dx(p)) {
            // dX with the WHOLE backward of the short-input layer 0 as its epilogue is the step's last GEMM: phase 2 (the caller
            // queues the next batch's assembly in front of it, as in front of a planes layer 0's dW)
            smallk_done = true;
            break;
        } else if ((p.fuse_ln_dx & 1) && B.kind != GTE_LAYER_SMALLK && gte_gemm_p3_nt_ln_bwd_supported(L.fin) &&
                   (L.fin % 16 == 0 || ldf(B) >= ((L.fin + 3) & ~(int64_t)3))) {      // (unaligned widths: padded rows)
            GTE_TRY(gte_gemm_p3_nt_ln_bwd(L.dzp, L.ldp_o, L.fout, L.qp, L.ldp_o, L.fout, L.wimg_bwd, L.ldp_wbwd, z_of(B), ldz_of(B), B.stats,
                                          B.gamma, B.beta, B.relu, (B.kind == GTE_LAYER_CACHED && B.dzp && !keep_dz0) ? nullptr : B.dy, ldf(B), B.dzp, B.ldp_o,
                                          B.ggamma, B.gbeta, B.gbias, n, L.fin, B.ws_ln, B.ws_ln_bytes, st));      // (cached input layer: dW reads
            ln_done = true;                                                                                      //  the image only)
        } else {
            GTE_TRY(gte_gemm_p3_nt(L.dzp, L.ldp_o, L.fout, L.qp, L.ldp_o, L.fout, L.wimg_bwd, L.ldp_wbwd, nullptr, 0, B.dy, ldf(B), n, L.fin, 0,
                                   0, st));
        }
    }
    return GTE_OK;
}

int backward_b(const gte_step_plan& p, void* st) {
    const int64_t n = p.n_nodes;
    const gte_step_layer& L = p.layer[0];
    if (smallk_in_dx(p)) {                             // layer 1's dX with layer 0's whole backward as its epilogue
        const gte_step_layer& U = p.layer[1];
        return gte_gemm_p3_nt_smallk_bwd(U.dzp, U.ldp_o, U.fout, U.qp, U.ldp_o, U.fout, U.wimg_bwd, U.ldp_wbwd, L.x, L.ldx, L.fin, L.ahn,
                                         L.fin, L.fin, L.W, 2 * L.fin, L.bias, L.gamma, L.beta, L.stats, L.relu, L.gW, 2 * L.fin, L.gbias,
                                         L.ggamma, L.gbeta, n, U.fin, L.ws_dw, L.ws_dw_bytes, st);
    }
    if (L.kind == GTE_LAYER_SMALLK && gte_sage_smallk_bwd_supported(2 * L.fin, L.fout))
        return gte_sage_smallk_bwd(L.dy, L.fout, L.x, L.ldx, L.fin, L.ahn, L.fin, L.fin, L.W, 2 * L.fin, L.bias, L.gamma, L.beta, L.stats,
                                   L.relu, L.gW, 2 * L.fin, L.gbias, L.ggamma, L.gbeta, n, L.fout, L.ws_dw, L.ws_dw_bytes, st);
    if (L.kind == GTE_LAYER_SMALLK)
        return gte_sage_linear_dw(L.dy, L.fout, L.x, L.ldx, L.fin, L.ahn, L.fin, L.fin, L.gW, 2 * L.fin, L.fout, n, L.ws_dw, L.ws_dw_bytes, st);
    if (L.kind == GTE_LAYER_CACHED && !L.h_rows)       // (the graph's own images: as the aggregate-first layer below)
        return gte_gemm_p3_tn(L.dzp, L.ldp_o, nullptr, 0, L.hp, L.ldp_h, L.ahnp, L.ldp_ahn, L.fin, L.gW, 2 * L.fin, L.fout, 2 * L.fin, n,
                              L.ws_dw, L.ws_dw_bytes, st);
    if (L.kind == GTE_LAYER_CACHED)                    // dW = [dz^T x | dz^T ahn], both operands resident behind the row map
        return gte_gemm_p3_tn_rows2(L.dzp, L.ldp_o, L.hp, L.ldp_h, L.ahnp, L.ldp_ahn, L.h_rows, L.n_res_rows, L.fin, L.gW, 2 * L.fin,
                                    L.fout, 2 * L.fin, n, L.ws_dw, L.ws_dw_bytes, st);
    if (L.kind == GTE_LAYER_AGGFIRST)                  // dW = [dz^T x | dz^T ahn]
        return gte_gemm_p3_tn(L.dzp, L.ldp_o, nullptr, 0, L.hp, L.ldp_h, L.ahnp, L.ldp_ahn, L.fin, L.gW, 2 * L.fin, L.fout, 2 * L.fin, n,
                              L.ws_dw, L.ws_dw_bytes, st);
    if (L.h_rows)
        return gte_gemm_p3_tn_rows(L.dzp, L.ldp_o, L.qp, L.ldp_o, L.hp, L.ldp_h, L.h_rows, L.n_res_rows, L.fin, L.gW, 2 * L.fin, L.fout,
                                   2 * L.fin, n, L.ws_dw, L.ws_dw_bytes, st);
    return gte_gemm_p3_tn(L.dzp, L.ldp_o, L.qp, L.ldp_o, L.hp, L.ldp_h, nullptr, 0, L.fin, L.gW, 2 * L.fin, L.fout, 2 * L.fin, n, L.ws_dw,
                          L.ws_dw_bytes, st);
}

int flush(const gte_step_plan& p, int* adam_fused) {
    if (adam_fused) *adam_fused = 0;
    if (p.param) {
        int fused = 0;
        const bool img = p.wimg_in_fold && p.n_wimg_descs > 0;
        const int rc = gte_fold_defer_flush_adam_images(p.param, p.grad, p.exp_avg, p.exp_avg_sq, p.n_param, p.hyper, p.step_counter, p.ticket,
                                                        img ? reinterpret_cast<const gte_p3_desc*>(p.wimg_descs) : nullptr,
                                                        img ? p.n_wimg_descs : 0, &fused);
        if (adam_fused) *adam_fused = fused;
        return rc;
    }
    return gte_fold_defer_flush();
}

int check_plan(const gte_step_plan& p) {
    if (p.n_hidden < 1 || p.n_hidden > 7 || p.n_nodes < 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_step: bad plan");
    for (int i = 0; i < p.n_hidden; ++i) {
        const gte_step_layer& L = p.layer[i];
        if (L.kind != GTE_LAYER_PLANES && L.kind != GTE_LAYER_SMALLK && L.kind != GTE_LAYER_AGGFIRST && L.kind != GTE_LAYER_CACHED)
            return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_step: layer kind");
        if (L.kind == GTE_LAYER_CACHED && (!L.hp || !L.ahnp || (L.h_rows && L.n_res_rows <= 0)))
            return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_step: a cached-aggregate layer needs the images of the input and of its aggregate");
        if (L.kind != GTE_LAYER_PLANES && i != 0)
            return gte::fail(GTE_ERR_UNSUPPORTED, "gcnsage_step: a short-input / aggregate-first layer must be layer 0");
        if (!L.gamma || !L.beta || !L.bias) return gte::fail(GTE_ERR_UNSUPPORTED, "gcnsage_step: hidden layers need bias and LayerNorm");
        if (L.ldf != 0 && (L.ldf < L.fout || L.ldf % 4 != 0)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_step: ldf < fout or not a multiple of 4");
        if (L.kind == GTE_LAYER_SMALLK && ldf(L) != L.fout) return gte::fail(GTE_ERR_UNSUPPORTED, "gcnsage_step: a short-input layer takes unpadded rows");
    }
    if (p.out_gemm && (p.n_classes > 16 || ld_lg(p) != 32 || !p.hp_out || !p.wimg_out_fwd || !p.wimg_out_bwd || !p.dlqp || p.tn != p.logits + 16 ||
                       p.q_out != p.dl + 16))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_step: the GEMM output layer needs n_classes <= 16, [n][32] logits / dl buffers and its images");
    if (p.n_nodes == 0) return gte::fail(GTE_ERR_UNSUPPORTED, "gcnsage_step: empty batch");
    return GTE_OK;
}

}  // namespace

extern "C" int gte_gcnsage_step(const gte_step_plan* plan, int phase, int* adam_fused, void* stream) {
    if (!plan) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_step: null plan");
    const gte_step_plan& p = *plan;
    if (phase < 0 || phase > 2) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_step: phase must be 0, 1 or 2");
    GTE_TRY(check_plan(p));
    if (phase != 2) {
        GTE_TRY(gte_gemm_set_tail_workspace(p.tail_ws, p.tail_ws ? p.tail_ws_bytes : 0));
        int rc = forward(p, stream);
        if (rc == GTE_OK) rc = gte_fold_defer_begin(stream);
        if (rc == GTE_OK) {
            rc = backward_a(p, stream);
            if (rc != GTE_OK) (void)gte_fold_defer_flush();        // leave no deferral open behind an error
        }
        if (rc != GTE_OK) { (void)gte_gemm_set_tail_workspace(nullptr, 0); return rc; }
        if (phase == 1) return GTE_OK;
    }
    int rc = backward_b(p, stream);
    if (rc != GTE_OK) { (void)gte_fold_defer_flush(); (void)gte_gemm_set_tail_workspace(nullptr, 0); return rc; }
    rc = flush(p, adam_fused);
    (void)gte_gemm_set_tail_workspace(nullptr, 0);
    return rc;
}

// The forward pass alone (model_predict.py:141-147 / the no_grad forward of model_train.py:349-353 of the reference): the
// same plan, the same kernels and buffers as the step's forward -- logits [n, C] land in plan->logits.  Labels, gradients,
// optimiser state and the backward workspaces of the plan are not touched.
extern "C" int gte_gcnsage_forward(const gte_step_plan* plan, void* stream) {
    if (!plan) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gcnsage_forward: null plan");
    const gte_step_plan& p = *plan;
    GTE_TRY(check_plan(p));
    GTE_TRY(gte_gemm_set_tail_workspace(p.tail_ws, p.tail_ws ? p.tail_ws_bytes : 0));
    const int64_t n = p.n_nodes, C = p.n_classes;
    int rc = forward_hidden(p, stream, true);
    // logits = h W_s^T + b + mean-aggregate(h W_n^T): the class-count-wide aggregation the step runs inside its loss kernel
    if (rc == GTE_OK) rc = forward_out_products(p, stream);
    if (rc == GTE_OK)
        rc = gte_spmm_csr_accumulate(p.indptr, p.indices, p.w_in, p.tn, ld_lg(p), p.logits, ld_lg(p), n, C, GTE_F32, GTE_REDUCE_MEAN, stream);
    (void)gte_gemm_set_tail_workspace(nullptr, 0);
    return rc;
}
