// One step of the short-input layer's backward (gte_sage_smallk_bwd; also the epilogue of gte_gemm_p3_nt_smallk_bwd): ROWS rows of
// one wave, lane l = output columns 4 l .. 4 l + 3.  z = [x | ahn] W^T + b is RECOMPUTED with the forward kernel's instruction
// sequence (sage_smallk_fwd_kernel: bit-identical z, the ReLU mask cannot flip), then the LayerNorm(+ReLU) backward of the rows
// (the arithmetic of ln_relu_bwd_vec_kernel) and dW[j + e][k] += dz[r][j + e] xin[r][k]; dz never leaves the registers.
//   xq: the rows' inputs in LDS ([row][Kp], columns K .. Kp-1 zero);  wl: W^T in LDS ([Kp][ns]) + this lane's column offset.
#pragma once
#include "gte_common.h"

#ifndef SKB_ABL
#define SKB_ABL 0
#endif

template <int KMAX, int ROWS>
__device__ __forceinline__ void gte_smallk_bwd_step(const float (&gy)[ROWS][4], const float (&mean)[ROWS], const float (&rstd)[ROWS],
                                                    const bool (&rok)[ROWS], bool ok, const float* xq, int Kp, const float* wl, int ns,
                                                    const float (&b4)[4], const float (&g4)[4], const float (&be4)[4], int relu,
                                                    float inv_n, float (&dw)[4][KMAX], float (&s_dg)[4], float (&s_db)[4],
                                                    float (&s_dbias)[4]) {
    float acc[ROWS][4];
#pragma unroll
    for (int u = 0; u < ROWS; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[u][e] = b4[e];
#pragma unroll
    for (int kc = 0; kc < KMAX; kc += 4) {
        if (kc >= Kp || (SKB_ABL & 1)) break;                  // uniform
        float4 w4[4], x4[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) x4[u] = *reinterpret_cast<const float4*>(xq + u * Kp + kc);    // broadcast
#pragma unroll
        for (int i = 0; i < 4; ++i) w4[i] = *reinterpret_cast<const float4*>(wl + (kc + i) * ns);
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            const float xk[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[u][0] = fmaf(xk[i], w4[i].x, acc[u][0]); acc[u][1] = fmaf(xk[i], w4[i].y, acc[u][1]);
                acc[u][2] = fmaf(xk[i], w4[i].z, acc[u][2]); acc[u][3] = fmaf(xk[i], w4[i].w, acc[u][3]);
            }
        }
    }
    float dz[ROWS][4];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
        float xh[4], g[4];
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh[e] = ok ? (acc[u][e] - mean[u]) * rstd[u] : 0.f;
            float gv = gy[u][e];
            if (relu && fmaf(xh[e], g4[e], be4[e]) <= 0.f) gv = 0.f;
            g[e] = gv;
            const float dxh = gv * g4[e];
            a += dxh;
            b = fmaf(dxh, xh[e], b);
        }
        const float c1 = gte_group_sum<64>(a) * inv_n, c2 = gte_group_sum<64>(b) * inv_n;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = rstd[u] * (g[e] * g4[e] - c1 - xh[e] * c2);
            dz[u][e] = (ok && rok[u]) ? d : 0.f;
            if (rok[u]) {                                       // wave-uniform
                s_dg[e] = fmaf(g[e], xh[e], s_dg[e]);
                s_db[e] += g[e];
                s_dbias[e] += ok ? d : 0.f;
            }
        }
    }
#pragma unroll
    for (int kc = 0; kc < KMAX; kc += 4) {
        if (kc < Kp && !(SKB_ABL & 2)) {                         // uniform
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const float4 x4 = *reinterpret_cast<const float4*>(xq + u * Kp + kc);             // broadcast
                const float xk[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dw[e][kc + i] = fmaf(dz[u][e], xk[i], dw[e][kc + i]);
            }
        }
    }
}
