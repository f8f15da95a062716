// Loss and optimiser of the train step (gfx950).
//
// gte_weighted_ce : nn.CrossEntropyLoss(weight)(logits, labels.long()) forward + backward and the
//                   accuracy count -- reference src/models/model_train.py:171,327-328.
// gte_adam_step   : torch.optim.Adam(lr, weight_decay).step() on one flat fp32 buffer --
//                   reference src/models/model_train.py:168,332.  L2-coupled decay (not AdamW).
//
// Both are HBM-bound and tiny next to the transforms (N x 9 logits; <= 4 M parameters).  The CE
// is deterministic: per-block partial sums in a fixed order, one reducing block, no atomics.
#include "gte_common.h"

#include <stdlib.h>
#include "fold_images.h"
#include "ce_fold.h"
#include "p3.h"

#include <stdint.h>

#include <math.h>

namespace {

using gte_ce::kCeBlock;
using gte_ce::label_of;
using gte_ce::ce_fold;
using gte_ce::ce_write_out3;

// pass 1: per-node nll / weight / correct, block-reduced into partial[block][3]
template <typename L>
__global__ void __launch_bounds__(kCeBlock)
ce_partial_kernel(const float* __restrict__ logits, int64_t ld, const L* __restrict__ labels,
                  const float* __restrict__ cw, int64_t n, int c, float* __restrict__ partial) {
    const int64_t i = (int64_t)blockIdx.x * kCeBlock + threadIdx.x;
    float loss = 0.f, wsum = 0.f, correct = 0.f;
    if (i < n) {
        const float* row = logits + i * ld;
        float m = row[0];
        int arg = 0;
        for (int j = 1; j < c; ++j) {
            const float v = row[j];
            if (v > m) { m = v; arg = j; }           // first maximum, as torch.argmax
        }
        float s = 0.f;
        for (int j = 0; j < c; ++j) s += expf(row[j] - m);
        const int y = label_of(labels, i);
        if (y >= 0 && y < c) {
            const float w = cw ? cw[y] : 1.0f;
            loss = w * (logf(s) + m - row[y]);
            wsum = w;
            correct = (arg == y) ? 1.f : 0.f;
        }
    }
    // wave reduce (64 lanes) then across the 4 waves through LDS, fixed order
    __shared__ float red[3][kCeBlock / gte::kWave];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        loss += __shfl_down(loss, off, 64);
        wsum += __shfl_down(wsum, off, 64);
        correct += __shfl_down(correct, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = loss; red[1][wave] = wsum; red[2][wave] = correct; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f, d = 0.f;
        for (int w = 0; w < kCeBlock / gte::kWave; ++w) { a += red[0][w]; b += red[1][w]; d += red[2][w]; }
        partial[(int64_t)blockIdx.x * 3 + 0] = a;
        partial[(int64_t)blockIdx.x * 3 + 1] = b;
        partial[(int64_t)blockIdx.x * 3 + 2] = d;
    }
}

// pass 2 (no gradient wanted): one block folds the partials -> out3 = {loss, sum w, #correct}
__global__ void __launch_bounds__(kCeBlock)
ce_final_kernel(const float* __restrict__ partial, int64_t nblocks, float* __restrict__ out3) {
    __shared__ double red[3][kCeBlock];
    ce_fold(partial, nblocks, red);
    if (threadIdx.x == 0) ce_write_out3(red, out3);
}

// pass 2+3: dlogits = grad_scale * w_y / sum_w * (softmax - onehot).  Every block repeats the (tiny) fold of the
// partials itself -- same order, same value everywhere -- instead of waiting for a one-block kernel in between;
// block 0 also writes out3.
template <typename L>
__global__ void __launch_bounds__(kCeBlock)
ce_grad_kernel(const float* __restrict__ logits, int64_t ld, const L* __restrict__ labels,
               const float* __restrict__ cw, int64_t n, int c, float grad_scale,
               const float* __restrict__ partial, int64_t nblocks, float* __restrict__ out3,
               float* __restrict__ dl, int64_t lddl) {
    __shared__ double red[3][kCeBlock];
    ce_fold(partial, nblocks, red);
    const float wsum = (float)red[1][0];
    if (blockIdx.x == 0 && threadIdx.x == 0) ce_write_out3(red, out3);
    const int64_t i = (int64_t)blockIdx.x * kCeBlock + threadIdx.x;
    if (i >= n) return;
    const float* row = logits + i * ld;
    float* drow = dl + i * lddl;
    const int y = label_of(labels, i);
    const bool ok = (y >= 0 && y < c && wsum > 0.f);
    const float w = ok ? (cw ? cw[y] : 1.0f) * grad_scale / wsum : 0.f;
    float m = row[0];
    for (int j = 1; j < c; ++j) m = fmaxf(m, row[j]);
    float s = 0.f;
    for (int j = 0; j < c; ++j) s += expf(row[j] - m);
    const float inv = 1.0f / s;
    for (int j = 0; j < c; ++j) {
        const float p = expf(row[j] - m) * inv;
        drow[j] = w * (p - (j == y ? 1.f : 0.f));
    }
}

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
            float grad_scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, grad_scale * g[i]);       // L2-coupled decay: grad + wd * param
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);       // exp_avg.lerp_(grad, 1-b1)
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// The same update with the hyper-parameters, the bias corrections and the step count read from DEVICE memory, so the
// launch can sit inside a HIP graph (the host-scalar version bakes lr and the bias corrections into the launch).
//   state = {lr, beta1, beta2, eps, weight_decay, grad_scale, bc1, sqrt(bc2)} for step t = *step_counter + 1.
// The block that finishes LAST (ticket counter) advances the state to step t + 1: every block has read the state before it
// takes its ticket, so the update cannot race with a reader.  bc in double like gte_adam_step's host side.
// IMG: behind its update every thread writes the three bf16 pieces of the new parameter values into the operand images that hold
// them (fold_images.h) -- the data-parallel step's Adam launch then leaves the next forward's weight images behind, as the
// fold + Adam launch of the one-GPU step does (no conversion launch in front of the forward)
template <bool IMG>
__global__ void __launch_bounds__(1024)
adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                int64_t n, float* __restrict__ state, int64_t* __restrict__ step_counter, unsigned* __restrict__ ticket,
                const gte::FoldImages img) {
    const gte::AdamCoef co = gte::adam_coef(state);
    const unsigned all = img.n >= 32 ? 0xffffffffu : ((1u << img.n) - 1u);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto update = [&](float& pi, float gi_raw, float& mi_io, float& vi_io) { gte::adam_update(co, pi, gi_raw, mi_io, vi_io); };
    // 16-byte accesses over the aligned body (the flat buffers come from one allocation each: 16-byte aligned bases),
    // scalar tail; the same per-element arithmetic as adam_kernel
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15) == 0 ? n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        update(pv.x, gv.x, mv.x, vv.x); update(pv.y, gv.y, mv.y, vv.y);
        update(pv.z, gv.z, mv.z, vv.z); update(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        reinterpret_cast<float4*>(p)[i] = pv;
        if constexpr (IMG) {
            gte::fold_write_images2(img, all, 4 * i, pv.x, pv.y);
            gte::fold_write_images2(img, all, 4 * i + 2, pv.z, pv.w);
        }
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        update(pi, g[i], mi, vi);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if constexpr (IMG) gte::fold_write_images(img, all, i, pi);
    }
    gte::adam_advance(state, reinterpret_cast<long long*>(step_counter), ticket);
}

}  // namespace

extern "C" int64_t gte_weighted_ce_workspace_bytes(int64_t n_nodes) {
    return gte::round_up(gte::ceil_div(n_nodes > 0 ? n_nodes : 1, kCeBlock) * 3 * (int64_t)sizeof(float), 256);
}

extern "C" int gte_weighted_ce(const float* logits, int64_t ld, const void* labels, int labels_f32,
                               const float* class_weight, int64_t n_nodes, int n_classes, float grad_scale,
                               float* dlogits, int64_t lddl, float* out3, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    if (n_nodes <= 0 || n_classes <= 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "weighted_ce: empty input");
    if (!logits || !labels || !out3 || !workspace) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "weighted_ce: null pointer");
    if (ld < n_classes || (dlogits && lddl < n_classes)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "weighted_ce: ld < n_classes");
    if (workspace_bytes < gte_weighted_ce_workspace_bytes(n_nodes))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "weighted_ce: workspace too small");
    hipStream_t s = gte::as_stream(stream);
    const int64_t nb = gte::ceil_div(n_nodes, kCeBlock);
    float* partial = reinterpret_cast<float*>(workspace);
    if (labels_f32)
        hipLaunchKernelGGL(ce_partial_kernel<float>, dim3((unsigned)nb), dim3(kCeBlock), 0, s, logits, ld,
                           (const float*)labels, class_weight, n_nodes, n_classes, partial);
    else
        hipLaunchKernelGGL(ce_partial_kernel<int64_t>, dim3((unsigned)nb), dim3(kCeBlock), 0, s, logits, ld,
                           (const int64_t*)labels, class_weight, n_nodes, n_classes, partial);
    if (dlogits) {
        if (labels_f32)
            hipLaunchKernelGGL(ce_grad_kernel<float>, dim3((unsigned)nb), dim3(kCeBlock), 0, s, logits, ld,
                               (const float*)labels, class_weight, n_nodes, n_classes, grad_scale, partial, nb, out3, dlogits,
                               lddl);
        else
            hipLaunchKernelGGL(ce_grad_kernel<int64_t>, dim3((unsigned)nb), dim3(kCeBlock), 0, s, logits, ld,
                               (const int64_t*)labels, class_weight, n_nodes, n_classes, grad_scale, partial, nb, out3, dlogits,
                               lddl);
    } else {
        hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(kCeBlock), 0, s, partial, nb, out3);
    }
    return gte::check_launch("weighted_ce");
}

// ---- column sums of a narrow matrix: out[c] = sum_r x[r][c], c < n <= 64 (the bias gradient of the output layer when its
// backward runs on the planes GEMMs: dbias = colsum(dlogits), models.py:101-103 through autograd) ----------------------------
// Block partials in a fixed order, folded by the step's deferred fold launch (or one tiny launch): deterministic, no atomics.
namespace {
constexpr int kColsumBlocks = 256;
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ x, long long ldx, int M, int n, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    if (lane < n)
        for (int r = blockIdx.x * 4 + wave; r < M; r += gridDim.x * 4) s += x[(long long)r * ldx + lane];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && lane < n) partial[(long long)blockIdx.x * n + lane] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ void __launch_bounds__(64)
colsum_fold_small_kernel(const float* __restrict__ partial, int nb, int n, float* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[(long long)b * n + c];
    out[c] = s;
}
int colsum_blocks(int64_t M) { return (int)(gte::ceil_div(M, 4) < kColsumBlocks ? gte::ceil_div(M, 4) : kColsumBlocks); }
}  // namespace

extern "C" int64_t gte_colsum_workspace_bytes(int64_t n_rows, int64_t n_cols) {
    return gte::round_up((int64_t)colsum_blocks(n_rows > 0 ? n_rows : 1) * (n_cols > 0 ? n_cols : 1) * 4, 256);
}

extern "C" int gte_colsum(const float* x, int64_t ldx, int64_t n_rows, int64_t n_cols, float* out, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    if (n_rows <= 0 || n_cols <= 0 || n_cols > 64 || n_rows > INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "colsum: needs 1 <= n_cols <= 64, n_rows >= 1");
    if (!x || !out || !workspace || ldx < n_cols) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "colsum: null pointer or ld < n_cols");
    if (workspace_bytes < gte_colsum_workspace_bytes(n_rows, n_cols)) return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "colsum: workspace too small");
    hipStream_t s = gte::as_stream(stream);
    const int nb = colsum_blocks(n_rows);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, (long long)ldx, (int)n_rows, (int)n_cols, part);
    if (!gte::defer_fold(part, n_cols, nb, 1, (int)n_cols, out, n_cols))
        hipLaunchKernelGGL(colsum_fold_small_kernel, dim3(1), dim3(64), 0, s, part, nb, (int)n_cols, out);
    return gte::check_launch("colsum");
}

// ---- the second half of the fused head on the GEMM output path ---------------------------------------------------------------
// gte_head_agg_ce leaves dl WITHOUT the 1 / sum(w) of the weighted cross-entropy (and q = A_w^T (norm dl) inherits that).  One
// launch: every block folds the CE partials for itself (fixed order: the same alpha everywhere), block 0 publishes out3; a thread
// per node writes alpha [dl | q] as ONE P3 image [n][32] (the operand of the dW / dh GEMMs of the output layer) and the block adds
// up alpha dl by columns (the bias gradient; block partials folded in order).  With the out-edge CSR (rindptr != NULL) the thread
// also forms its node's q = A_w^T (norm dl) itself from the dl rows of its out-neighbours (the q columns of dlq are not read).
// Replaces the loss-gradient launch, the 9-wide transpose aggregation, the image conversion and the column-sum launch.
namespace {
constexpr int DLQ_ROWS = 256;
__global__ void __launch_bounds__(DLQ_ROWS)
head_dlq_finish_kernel(const float* __restrict__ dlq, long long ld, int n, int C, const float* __restrict__ ce_partial, long long ce_blocks,
                       float grad_scale, float* __restrict__ out3, char* __restrict__ dlqp, long long ldp, float* __restrict__ bias_part,
                       const int32_t* __restrict__ rindptr, const int32_t* __restrict__ rindices, const float* __restrict__ w_out) {
    __shared__ double ce_red[3][kCeBlock];
    static_assert(kCeBlock == DLQ_ROWS, "one fold per block");
    ce_fold(ce_partial, ce_blocks, ce_red);
    const float wsum = (float)ce_red[1][0];
    const float alpha = wsum > 0.f ? grad_scale / wsum : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && out3) ce_write_out3(ce_red, out3);
    const int row = blockIdx.x * DLQ_ROWS + threadIdx.x;
    float v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = 0.f;
    if (row < n) {
        const float4* src = reinterpret_cast<const float4*>(dlq + (long long)row * ld);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 t = src[k];
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
        if (rindptr) {
            // q = A_w^T (norm dl) of this node, here instead of by a 9-wide aggregation launch: the row's out-edges in CSR order,
            // fmaf(w, dl[u], acc) -- the arithmetic of spmm_csr_kernel
#pragma unroll
            for (int c = 16; c < 32; ++c) v[c] = 0.f;
            const int e1 = rindptr[row + 1];
            for (int e = rindptr[row]; e < e1; ++e) {
                const float w = w_out ? w_out[e] : 1.f;
                const float4* ur = reinterpret_cast<const float4*>(dlq + (long long)rindices[e] * ld);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (4 * k < C) {                                   // uniform
                        const float4 t = ur[k];
                        v[16 + 4 * k] = fmaf(w, t.x, v[16 + 4 * k]); v[17 + 4 * k] = fmaf(w, t.y, v[17 + 4 * k]);
                        v[18 + 4 * k] = fmaf(w, t.z, v[18 + 4 * k]); v[19 + 4 * k] = fmaf(w, t.w, v[19 + 4 * k]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = (c & 15) < C ? v[c] * alpha : 0.f;      // (columns C .. 15 of a half are not data)
        char* img = dlqp + (long long)row * ldp;
#pragma unroll
        for (int k = 0; k < 8; ++k) p3::store4(img, 4 * k, v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
    // column sums of alpha dl over the block's rows: waves by shuffles, the four waves in order
    __shared__ float cs[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float s_ = gte_group_sum<64>(v[c]);
        if (lane == 0) cs[wave][c] = s_;
    }
    __syncthreads();
    if (threadIdx.x < 16) bias_part[(long long)blockIdx.x * 16 + threadIdx.x] = (cs[0][threadIdx.x] + cs[1][threadIdx.x]) + (cs[2][threadIdx.x] + cs[3][threadIdx.x]);
}
__global__ void __launch_bounds__(64)
head_bias_fold_kernel(const float* __restrict__ part, int nb, int C, float* __restrict__ out) {
    if ((int)threadIdx.x >= C) return;
    float s_ = 0.f;
    for (int k = 0; k < nb; ++k) s_ += part[(long long)k * 16 + threadIdx.x];
    out[threadIdx.x] = s_;
}
}  // namespace

extern "C" int64_t gte_head_dlq_finish_workspace_bytes(int64_t n_nodes) {
    return gte::round_up(gte::ceil_div(n_nodes > 0 ? n_nodes : 1, DLQ_ROWS) * 16 * 4, 256);
}

extern "C" int gte_head_dlq_finish(const int32_t* rindptr, const int32_t* rindices, const float* w_out, const float* dlq, int64_t lddlq,
                                   int64_t n_nodes, int64_t n_classes, const void* ce_partial, float grad_scale, float* out3, void* dlqp3,
                                   int64_t ldp, float* gbias, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_nodes <= 0 || n_nodes > INT32_MAX || n_classes < 1 || n_classes > 16)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "head_dlq_finish: needs 1 <= n_classes <= 16 and n_nodes >= 1");
    if (!dlq || !ce_partial || !out3 || !dlqp3 || !gbias || !workspace) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "head_dlq_finish: null pointer");
    if (lddlq < 32 || lddlq % 4 != 0 || (reinterpret_cast<uintptr_t>(dlq) & 15) != 0 || ldp < 192 || ldp % 16 != 0)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "head_dlq_finish: dlq is [n][>= 32] floats, 16-byte aligned rows; the image two blocks wide");
    if (workspace_bytes < gte_head_dlq_finish_workspace_bytes(n_nodes))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "head_dlq_finish: workspace too small");
    hipStream_t s = gte::as_stream(stream);
    const int nb = (int)gte::ceil_div(n_nodes, DLQ_ROWS);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(head_dlq_finish_kernel, dim3((unsigned)nb), dim3(DLQ_ROWS), 0, s, dlq, (long long)lddlq, (int)n_nodes, (int)n_classes,
                       reinterpret_cast<const float*>(ce_partial), (long long)gte::ceil_div(n_nodes, 64), grad_scale, out3,
                       reinterpret_cast<char*>(dlqp3), (long long)ldp, part, rindptr, rindices, w_out);
    if (!gte::defer_fold(part, 16, nb, 1, (int)n_classes, gbias, n_classes))
        hipLaunchKernelGGL(head_bias_fold_kernel, dim3(1), dim3(64), 0, s, part, nb, (int)n_classes, gbias);
    return gte::check_launch("head_dlq_finish");
}

extern "C" int gte_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                             float grad_scale, void* stream) {
    if (n < 0 || step < 1) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "adam_step: n < 0 or step < 1");
    if (n == 0) return GTE_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "adam_step: null pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t blocks = gte::ceil_div(n, 256) < 2048 ? gte::ceil_div(n, 256) : 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, gte::as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return gte::check_launch("adam_step");
}

extern "C" int64_t gte_adam_ticket_bytes(void) { return (int64_t)(1 + gte::kTicketShards) * gte::kTicketStride * 4; }

extern "C" int gte_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                 int64_t* step_counter, unsigned* ticket, void* stream) {
    if (n <= 0) return n == 0 ? gte::fail(GTE_ERR_INVALID_ARGUMENT, "adam_step_dev: n == 0 (nothing would advance the step)")
                              : gte::fail(GTE_ERR_INVALID_ARGUMENT, "adam_step_dev: n < 0");
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || !step_counter || !ticket)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "adam_step_dev: null pointer");
    // one ticket atomic per block on a single address: 2048 blocks spent 25 us queueing on it; 256 blocks (a grid-stride
    // loop over ~2300 elements per block at 590 k parameters) keep the 16 MB update at ~5 us
    // 1024 threads per block: at 590 k parameters every thread updates ONE 16-byte group (a 256-thread block looped 2-3
    // times through dependent load -> store chains: 10 us)
    const int64_t blocks = gte::ceil_div(n, 4096) < 256 ? gte::ceil_div(n, 4096) : 256;
    gte::FoldImages none;
    none.n = 0;
    hipLaunchKernelGGL(adam_dev_kernel<false>, dim3((unsigned)blocks), dim3(1024), 0, gte::as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, n, state, step_counter, ticket, none);
    return gte::check_launch("adam_step_dev");
}

// gte_adam_step_dev that also writes the P3 images of the UPDATED parameters (gte_p3_desc[]: sub-matrices of `param`, as
// gte_fold_defer_flush_adam_images takes them): the optimiser launch of the data-parallel step -- behind the gradient all-reduce,
// where the fold launch cannot apply the update -- leaves the next forward's weight images behind.  *wrote = 1 when it did (a list
// the launch cannot carry -- more than 12 images -- is skipped: the caller converts as before).
extern "C" int gte_adam_step_dev_images(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                        int64_t* step_counter, unsigned* ticket, const gte_p3_desc* images, int n_images, int* wrote,
                                        void* stream) {
    if (wrote) *wrote = 0;
    if (n <= 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "adam_step_dev_images: n <= 0");
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || !step_counter || !ticket || !wrote)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "adam_step_dev_images: null pointer");
    gte::FoldImages fi;
    const int rc = gte::make_fold_images(param, n, images, n_images, fi, "adam_step_dev_images");
    if (rc != GTE_OK) return rc;
    // (round 5: workgroups of 256 -- one 16-byte group per thread up to 2 048 workgroups -- instead of <= 256 of 1 024: the launch is a
    // chain of dependent memory round trips plus scattered 2-byte image stores, more workgroups in flight shorten it)
    static const int bt = GTE_MEASURE_INT("GTE_ADAM_DEV_BLOCK", 256);      // (1024 = round 4)
    const int threads = bt == 1024 ? 1024 : 256;
    const int64_t want = gte::ceil_div(n, 4 * (int64_t)threads), cap = threads == 1024 ? 256 : 2048;
    const int64_t blocks = want < cap ? want : cap;
    if (fi.n > 0)
        hipLaunchKernelGGL(adam_dev_kernel<true>, dim3((unsigned)blocks), dim3(threads), 0, gte::as_stream(stream), param, grad, exp_avg,
                           exp_avg_sq, n, state, step_counter, ticket, fi);
    else
        hipLaunchKernelGGL(adam_dev_kernel<false>, dim3((unsigned)blocks), dim3(threads), 0, gte::as_stream(stream), param, grad, exp_avg,
                           exp_avg_sq, n, state, step_counter, ticket, fi);
    *wrote = fi.n > 0 ? 1 : 0;
    return gte::check_launch("adam_step_dev_images");
}
