// Backward of the short-input layer (BBOX features: 13 + 13 inputs) in ONE pass  --  gte_sage_smallk_bwd.
//
// Replaces, for layer 0 of GcnSAGE when 2 * in_feats <= 28 (models.py:63-66 backward: nn.LayerNorm / ReLU / nn.Linear autograd):
//   gte_ln_relu_bwd   (reads dy and the saved pre-LayerNorm z, writes dz)             75 MB at 24.5 k x 256, 22 us
//   gte_sage_linear_dw (dW = dz^T [x | ahn], a K = 24.5 k GEMM with 26 output columns)  25 MB,               16 us
// and lets the forward drop its 25 MB store of z: the layer's input is 26 floats per row, so z is RECOMPUTED here with the
// forward kernel's own instruction sequence (bit-identical z: the ReLU mask cannot flip), dz never leaves the registers
// (the input layer has no dX), and dW = sum_r dz[r, :]^T xin[r, :] is accumulated per lane (4 columns x K).
// Traffic: dy (25 MB) + inputs (2.5 MB) + per-workgroup partial sums.
//
// Layout = sage_smallk_fwd_kernel's: a workgroup owns 64 consecutive rows, W^T and the rows' inputs sit in LDS, lane l owns
// output columns 4 l .. 4 l + 3 of four rows per wave step.
#include "gte_common.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

#ifndef SKB_ABL
#define SKB_ABL 0         // measurement builds: 1 no z recomputation, 2 no dW accumulation, 4 no reduction epilogue, 8 no dy loads
#endif
#include "smallk_step.h"
constexpr int SKB_KLIMIT = 28;            // k1 + k2 (rounded up to 4) the per-lane dW accumulators cover (4 x 28 registers)
constexpr int SKB_ROWS = 2;               // rows per wave step (four in the forward kernel: the per-row FMA chains are the same)
constexpr int SKB_BLOCK_ROWS = 64;        // rows per workgroup pass: 4 waves x 8 steps x 2 rows

struct f4v { float x, y, z, w; };

__device__ __forceinline__ float wave_sum64(float v) { return gte_group_sum<64>(v); }

template <int SKB_KMAX>
__global__ void __launch_bounds__(256, 2)
sage_smallk_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ a1, int64_t lda1, int k1,
                       const float* __restrict__ a2, int64_t lda2, int k2, const float* __restrict__ W, int64_t ldw,
                       const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                       const float* __restrict__ stats, int relu, float* __restrict__ part_dw, float* __restrict__ part_cs, int M,
                       int n, int xs_floats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int K = k1 + k2, Kp = (K + 3) & ~3;
    const int ns = n;
    float* Wt = sm;                        // [Kp][ns] of [SKB_KMAX][ns], rows K .. Kp-1 zero; after the row loop: the reduction buffer
    float* xs = sm + SKB_KMAX * ns;        // [64][Kp], columns K .. Kp-1 zero
    {
        float wv[SKB_KMAX];
        const float* wr = W + (int64_t)min((int)threadIdx.x, n - 1) * ldw;
#pragma unroll
        for (int k = 0; k < SKB_KMAX; ++k) wv[k] = k < K ? wr[k] : 0.f;
        if ((int)threadIdx.x < n) {
#pragma unroll
            for (int k = 0; k < SKB_KMAX; ++k) if (k < Kp) Wt[k * ns + threadIdx.x] = wv[k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = 4 * lane;
    const bool ok = j < n;
    float b4[4] = {0.f, 0.f, 0.f, 0.f}, g4[4] = {0.f, 0.f, 0.f, 0.f}, be4[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { b4[e] = bias[j + e]; g4[e] = gamma[j + e]; be4[e] = beta[j + e]; }
    }
    const float inv_n = 1.0f / (float)n;
    const float* wl = Wt + (ok ? j : 0);
    float dw[4][SKB_KMAX];
    float s_dg[4] = {0.f, 0.f, 0.f, 0.f}, s_db[4] = {0.f, 0.f, 0.f, 0.f}, s_dbias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < SKB_KMAX; ++k) dw[e][k] = 0.f;

    const int rl = threadIdx.x >> 2, part = threadIdx.x & 3;           // input fill: 4 threads per row
    for (int brow0 = blockIdx.x * SKB_BLOCK_ROWS; brow0 < M; brow0 += gridDim.x * SKB_BLOCK_ROWS) {
        {
            float xv[SKB_KMAX / 4];
            const int r = min(brow0 + rl, M - 1);
            const float* r1 = a1 + (int64_t)r * lda1;
            const float* r2 = a2 ? a2 + (int64_t)r * lda2 : a1;
#pragma unroll
            for (int i = 0; i < SKB_KMAX / 4; ++i) {
                const int k = part + 4 * i;
                xv[i] = 0.f;
                if (k < k1) xv[i] = r1[k];
                else if (k < K) xv[i] = r2[k - k1];
            }
            __syncthreads();               // the previous pass' readers of xs are done (first pass: nothing pending)
#pragma unroll
            for (int i = 0; i < SKB_KMAX / 4; ++i) {
                const int k = part + 4 * i;
                if (k < Kp) xs[rl * Kp + k] = xv[i];
            }
        }
        __syncthreads();
        constexpr int STEPS = SKB_BLOCK_ROWS / (4 * SKB_ROWS);
        // dy of a step's rows and their statistics (rows past M clamped: their results are masked below)
        auto load_rows = [&](int step, float (&gy_)[SKB_ROWS][4], float (&mean_)[SKB_ROWS], float (&rstd_)[SKB_ROWS]) {
            const int r0 = brow0 + (step * 4 + wave) * SKB_ROWS;
#pragma unroll
            for (int u = 0; u < SKB_ROWS; ++u) {
                const int r = min(r0 + u, M - 1);
                f4v t{0.f, 0.f, 0.f, 0.f};
                if (ok && !(SKB_ABL & 8)) t = *reinterpret_cast<const f4v*>(dy + (int64_t)r * lddy + j);
                gy_[u][0] = t.x; gy_[u][1] = t.y; gy_[u][2] = t.z; gy_[u][3] = t.w;
                mean_[u] = stats[r]; rstd_[u] = stats[M + r];
            }
        };
        float gy[SKB_ROWS][4], mean[SKB_ROWS], rstd[SKB_ROWS];
        load_rows(0, gy, mean, rstd);
        for (int step = 0; step < STEPS; ++step) {
            const int rl0 = (step * 4 + wave) * SKB_ROWS;
            const int row0 = brow0 + rl0;
            if (row0 >= M) break;                                      // wave-uniform
            // the NEXT step's rows are requested now: a step is one dependent chain (loads -> z -> row sums -> dW), and with
            // two waves per SIMD nothing else hides the memory latency
            float gyn[SKB_ROWS][4], meann[SKB_ROWS], rstdn[SKB_ROWS];
            load_rows(step + 1 < STEPS ? step + 1 : step, gyn, meann, rstdn);
            bool rok[SKB_ROWS];
#pragma unroll
            for (int u = 0; u < SKB_ROWS; ++u) rok[u] = row0 + u < M;   // wave-uniform
            gte_smallk_bwd_step<SKB_KMAX, SKB_ROWS>(gy, mean, rstd, rok, ok, xs + rl0 * Kp, Kp, wl, ns, b4, g4, be4, relu, inv_n, dw, s_dg,
                                                    s_db, s_dbias);
#pragma unroll
            for (int u = 0; u < SKB_ROWS; ++u) {
                mean[u] = meann[u]; rstd[u] = rstdn[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) gy[u][e] = gyn[u][e];
            }
        }
    }
    // The four waves' partial sums through LDS (Wt is dead now: [SKB_KMAX][ns] floats), a quarter of the k range per round:
    // every wave stores its [KC][4 columns] slices, then thread t adds the four waves' values of column t in wave order and
    // puts them into the workgroup's [n][K] result IN LDS; global stores come last, coalesced, with no barrier behind them
    // (a barrier drains the store queue first: four rounds of scattered 4-byte stores cost 19 of the kernel's 41 us).
    constexpr int KC = SKB_KMAX / 4;
    float* red = sm;                       // [4 waves][KC][ns]
    float* obuf = xs + SKB_BLOCK_ROWS * Kp;    // [n][K]: the layout of dW (ld = K)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c * KC < Kp && !(SKB_ABL & (4 | 32))) {   // uniform
            if (ok) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    const int k = c * KC + kk;
                    float4 v; v.x = dw[0][k]; v.y = dw[1][k]; v.z = dw[2][k]; v.w = dw[3][k];
                    *reinterpret_cast<float4*>(red + (wave * KC + kk) * ns + j) = v;
                }
            }
            __syncthreads();
            if ((int)threadIdx.x < n) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    const int k = c * KC + kk;
                    if (k < K) {
                        const float* q = red + kk * ns + threadIdx.x;
                        obuf[threadIdx.x * K + k] = ((q[0] + q[KC * ns]) + q[2 * KC * ns]) + q[3 * KC * ns];
                    }
                }
            }
            __syncthreads();
        }
    }
    if (ok) {
        float4 v;
        v.x = s_dg[0]; v.y = s_dg[1]; v.z = s_dg[2]; v.w = s_dg[3];
        *reinterpret_cast<float4*>(red + (wave * 3 + 0) * ns + j) = v;
        v.x = s_db[0]; v.y = s_db[1]; v.z = s_db[2]; v.w = s_db[3];
        *reinterpret_cast<float4*>(red + (wave * 3 + 1) * ns + j) = v;
        v.x = s_dbias[0]; v.y = s_dbias[1]; v.z = s_dbias[2]; v.w = s_dbias[3];
        *reinterpret_cast<float4*>(red + (wave * 3 + 2) * ns + j) = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < n) {
        float* pc = part_cs + (int64_t)blockIdx.x * 3 * n + threadIdx.x;            // [block][3][n]
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
            const float* q = red + q3 * ns + threadIdx.x;
            pc[q3 * n] = ((q[0] + q[3 * ns]) + q[6 * ns]) + q[9 * ns];
        }
    }
    if (!(SKB_ABL & (4 | 64))) {
        float* pd = part_dw + (int64_t)blockIdx.x * n * K;                          // [block][n][K]
        for (int e = threadIdx.x; e < n * K; e += 256) pd[e] = obuf[e];
    }
}

// out[r * ld + c] = sum_k part[k * stride + r * rowlen + c]   (only when no fold deferral is open)
__global__ void __launch_bounds__(256)
smallk_fold_kernel(const float* __restrict__ part, int64_t stride, int count, int rows, int rowlen, float* __restrict__ out, int64_t ld) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)rows * rowlen) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < count; k += 4) {
        s0 += part[k * stride + e]; s1 += part[(k + 1) * stride + e]; s2 += part[(k + 2) * stride + e]; s3 += part[(k + 3) * stride + e];
    }
    for (; k < count; ++k) s0 += part[k * stride + e];
    const int64_t r = e / rowlen;
    out[r * ld + (e - r * rowlen)] = (s0 + s1) + (s2 + s3);
}

int skb_blocks(int64_t M) {
    const int64_t b = gte::ceil_div(M, SKB_BLOCK_ROWS);
    return (int)(b < 512 ? b : 512);
}

bool skb_supported(int64_t K, int64_t n_out) {
    static const bool off = GTE_MEASURE_OFF("GTE_SMALLK_BWD");
    return !off && K >= 1 && K <= SKB_KLIMIT && n_out % 4 == 0 && n_out >= 4 && n_out <= 256;
}

}  // namespace

extern "C" int gte_sage_smallk_bwd_supported(int64_t k_total, int64_t n_out) { return skb_supported(k_total, n_out) ? 1 : 0; }

extern "C" int64_t gte_sage_smallk_bwd_workspace_bytes(int64_t n_nodes, int64_t k_total, int64_t n_out) {
    const int64_t nb = skb_blocks(n_nodes > 0 ? n_nodes : 1);
    return gte::round_up(nb * (n_out > 0 ? n_out : 1) * ((k_total > 0 ? k_total : 1) + 3) * 4, 256);
}

extern "C" int gte_sage_smallk_bwd(const float* dy, int64_t lddy, const float* a1, int64_t lda1, int64_t k1, const float* a2,
                                   int64_t lda2, int64_t k2, const float* W, int64_t ldw, const float* bias, const float* gamma,
                                   const float* beta, const float* stats, int relu, float* dW, int64_t lddw, float* dbias,
                                   float* dgamma, float* dbeta, int64_t n_nodes, int64_t n_out, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    if (n_nodes < 0 || n_out <= 0 || k1 <= 0 || k2 < 0 || n_nodes > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_smallk_bwd: bad sizes");
    const int64_t K = k1 + k2;
    if (!skb_supported(K, n_out))
        return gte::fail(GTE_ERR_UNSUPPORTED, "sage_smallk_bwd: needs k1 + k2 <= %d, n_out %% 4 == 0, n_out <= 256", SKB_KLIMIT);
    if (n_nodes == 0) return GTE_OK;
    if (!dy || !a1 || (k2 > 0 && !a2) || !W || !bias || !gamma || !beta || !stats || !dW || !workspace)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_smallk_bwd: null pointer (the layer needs bias and LayerNorm)");
    if (lddy < n_out || lda1 < k1 || (k2 > 0 && lda2 < k2) || ldw < K || lddw < K)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_smallk_bwd: leading dimension too small");
    if ((reinterpret_cast<uintptr_t>(dy) & 15) || lddy % 4 != 0)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "sage_smallk_bwd: dy rows must be 16-byte aligned");
    const int64_t need = gte_sage_smallk_bwd_workspace_bytes(n_nodes, K, n_out);
    if (workspace_bytes < need)
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "sage_smallk_bwd: needs %lld workspace bytes, got %lld", (long long)need,
                         (long long)workspace_bytes);
    hipStream_t s = gte::as_stream(stream);
    const int nb = skb_blocks(n_nodes);
    const int64_t kp = (K + 3) & ~(int64_t)3;
    float* part_dw = reinterpret_cast<float*>(workspace);               // [nb][n_out][K]
    float* part_cs = part_dw + (int64_t)nb * n_out * K;                 // [nb][3][n_out]
    const int64_t xs_floats = SKB_BLOCK_ROWS * kp + n_out * K;          // the rows' inputs + the workgroup's [n_out][K] result
#define GTE_SKB(KM)                                                                                                              \
    hipLaunchKernelGGL((sage_smallk_bwd_kernel<KM>), dim3((unsigned)nb), dim3(256), (size_t)(KM * n_out + xs_floats) * sizeof(float), s, \
                       dy, lddy, a1, lda1, (int)k1, a2, lda2, (int)k2, W, ldw, bias, gamma, beta, stats, relu, part_dw, part_cs,      \
                       (int)n_nodes, (int)n_out, (int)xs_floats)
    if (kp <= 16) GTE_SKB(16); else GTE_SKB(28);
#undef GTE_SKB
    int rc = gte::check_launch("sage_smallk_bwd");
    if (rc != GTE_OK) return rc;
    // the partial sums join the step's fold batch (gte_fold_defer_begin) or are folded here
    if (gte::defer_fold(part_dw, n_out * K, nb, (int)n_out, (int)K, dW, lddw)) {
        gte::defer_fold(part_cs, 3 * n_out, nb, 1, (int)n_out, dgamma, n_out);
        gte::defer_fold(part_cs + n_out, 3 * n_out, nb, 1, (int)n_out, dbeta, n_out);
        gte::defer_fold(part_cs + 2 * n_out, 3 * n_out, nb, 1, (int)n_out, dbias, n_out);
        return GTE_OK;
    }
    hipLaunchKernelGGL(smallk_fold_kernel, dim3((unsigned)gte::ceil_div(n_out * K, 256)), dim3(256), 0, s, part_dw, n_out * K, nb,
                       (int)n_out, (int)K, dW, lddw);
    float* outs[3] = {dgamma, dbeta, dbias};
    for (int i = 0; i < 3; ++i)
        if (outs[i])
            hipLaunchKernelGGL(smallk_fold_kernel, dim3((unsigned)gte::ceil_div(n_out, 256)), dim3(256), 0, s, part_cs + i * n_out,
                               3 * n_out, nb, 1, (int)n_out, outs[i], n_out);
    return gte::check_launch("sage_smallk_bwd fold");
}
