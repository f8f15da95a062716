// Shared host-side helpers for libgte_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/gte.h"

namespace gte {

// Thread-local last-error text (gte_last_error()).  No exceptions cross the ABI.
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GTE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GTE_OK;
}

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kNumXcd = 8;         // MI355X: 8 XCDs, one L2 each

struct DeviceProps {
    int cus;
    int lds_bytes;
    char arch[64];
};
const DeviceProps& device_props();

// Deferred folds (gte_fold_defer_begin / _flush): while a deferral is open on this thread, a producer that would launch
// its own partial-sum fold kernel queues the fold instead; the flush runs every queued fold in ONE launch.
//   dst[r * ld + c] = sum_{k < count} src[k * stride + r * rowlen + c]   (fixed order: deterministic)
// Returns false when no deferral is open (the producer launches its own fold as usual).
bool defer_fold(const float* src, int64_t stride, int count, int rows, int rowlen, float* dst, int64_t ld);

// Caller-provided scratch for the GEMM tail split (gte_gemm_set_tail_workspace), thread-local; {nullptr, 0} when unset.
struct TailWorkspace { float* ptr; int64_t bytes; };
TailWorkspace tail_workspace();

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Measurement switches.  The shipped library reads ONE environment variable (GTE_GEMM_MODE, documented in include/gte.h); every
// other switch -- forced tile / block counts, kernel paths turned off for A/B runs -- exists only in the measurement build
// (-DGTE_MEASURE: libgte_hip_measure.so of the Makefile, which tests/test_gpu_variants.py and the scripts under profiles/ load
// through GTE_LIB_PATH).  In the shipped build the macros below are their default: no getenv, no name string in the binary.
#ifdef GTE_MEASURE
#define GTE_MEASURE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define GTE_MEASURE_OFF(name) (getenv(name) && getenv(name)[0] == '0')       /* true: the switch turns its path OFF */
#else
#define GTE_MEASURE_INT(name, dflt) (dflt)
#define GTE_MEASURE_OFF(name) (false)
#endif
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace gte

// ---- Adam on the device-resident optimiser state (shared by adam_dev_kernel and the fold kernel's optimiser tail) ----
// state = {lr, b1, b2, eps, wd, grad_scale, bc1, sqrt(bc2)}; torch.optim.Adam's update with L2 weight decay.
namespace gte {
struct AdamCoef { float lr, b1, b2, eps, wd, grad_scale, bc1, bc2_sqrt; };
__device__ __forceinline__ AdamCoef adam_coef(const float* __restrict__ state) {
    return AdamCoef{state[0], state[1], state[2], state[3], state[4], state[5], state[6], state[7]};
}
__device__ __forceinline__ void adam_update(const AdamCoef& c, float& pi, float gi_raw, float& mi_io, float& vi_io) {
    const float gi = fmaf(c.wd, pi, c.grad_scale * gi_raw);
    const float mi = fmaf(c.b1, mi_io, (1.f - c.b1) * gi);
    const float vi = fmaf(c.b2, vi_io, (1.f - c.b2) * gi * gi);
    mi_io = mi;
    vi_io = vi;
    const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
    pi = pi - (c.lr / c.bc1) * (mi / denom);
}
// End of a launch that applied one optimiser step: the LAST workgroup to finish advances the step count and the bias
// corrections for the next step (every workgroup has read `state` before it takes its ticket).  bc in double.
// Tickets are SHARDED: a device-scope atomic on one address retires every ~12 ns, and the fold + Adam launch has 1 000 - 1 600
// workgroups that finish together (r03: 13 - 20 us of a 25 - 32 us launch were that queue).  Workgroup b takes a ticket of
// shard b % 32 (its own 128-byte line); the workgroup that completes a shard takes one of the <= 32 top-level tickets.
// `ticket` = gte_adam_ticket_bytes() bytes, zero-initialised, returned to zero.
constexpr int kTicketShards = 32, kTicketStride = 32;        // (words: 128 bytes between counters)
__device__ __forceinline__ void adam_advance(float* __restrict__ state, long long* __restrict__ step_counter,
                                             unsigned* __restrict__ ticket) {
    __syncthreads();                                         // the whole block is done with `state`
    if (threadIdx.x == 0) {
        const unsigned nsh = gridDim.x < (unsigned)kTicketShards ? gridDim.x : (unsigned)kTicketShards;
        const unsigned sh = blockIdx.x % nsh;
        const unsigned members = (gridDim.x - sh + nsh - 1) / nsh;
        unsigned* mine = ticket + (1 + sh) * kTicketStride;
        if (atomicAdd(mine, 1u) != members - 1) return;
        *mine = 0;
        const unsigned t = atomicAdd(ticket, 1u);
        if (t == nsh - 1) {
            *ticket = 0;
            const long long done = *step_counter + 1;
            *step_counter = done;
            const double tn = (double)(done + 1);
            state[6] = (float)(1.0 - pow((double)state[1], tn));
            state[7] = (float)sqrt(1.0 - pow((double)state[2], tn));
        }
    }
}
}  // namespace gte

// XCD-aware block remap (device).  Workgroups are dealt round-robin over the 8 XCDs, so
// blocks b and b+8 share an L2.  Give each XCD a contiguous range of logical blocks so that
// neighbouring rows/tiles (which share gathered source rows / operand panels) hit one L2.
// Bijective for any grid size.  Speed only -- never correctness.
__device__ __forceinline__ unsigned gte_xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned xcd = bid % gte::kNumXcd;
    const unsigned q = nblocks / gte::kNumXcd, r = nblocks % gte::kNumXcd;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / gte::kNumXcd;
}

// ---- lane-group sums on DPP modifiers (device) ------------------------------------------------------------------
// Inside a row of 16 lanes the butterfly runs on DPP controls: xor 1 / xor 2 as quad permutes; once quads / octets are
// uniform, the half-row and row mirrors act as xor 4 / xor 8.  Four VALU adds, no LDS-pipe instruction -- __shfl_xor
// compiles to ds_bpermute_b32, one LDS round trip per level.  Result in every lane of the group.
#define GTE_DPP_ADD(v, ctrl) \
    (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))

template <int G>                       // G = 2 .. 64 lanes per group, power of two, groups aligned to G
__device__ __forceinline__ float gte_group_sum(float v) {
    if constexpr (G >= 2) GTE_DPP_ADD(v, 0xB1);          // quad_perm [1,0,3,2]: lane ^ 1
    if constexpr (G >= 4) GTE_DPP_ADD(v, 0x4E);          // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (G >= 8) GTE_DPP_ADD(v, 0x141);         // row_half_mirror (quads uniform: == lane ^ 4)
    if constexpr (G >= 16) GTE_DPP_ADD(v, 0x140);        // row_mirror (octets uniform: == lane ^ 8)
    if constexpr (G == 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G == 64) {                             // four row sums through the scalar unit
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
        v = (r0 + r1) + (r2 + r3);
    }
    return v;
}

// value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float gte_quad_swap1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

// value of lane T of every quad, in all four lanes of the quad (DPP quad_perm broadcast; T is a compile-time constant)
template <int T>
__device__ __forceinline__ int gte_quad_bcast(int v) {
    return __builtin_amdgcn_update_dpp(0, v, T | (T << 2) | (T << 4) | (T << 6), 0xf, 0xf, true);
}
template <int T>
__device__ __forceinline__ float gte_quad_bcast(float v) {
    return __builtin_bit_cast(float, gte_quad_bcast<T>(__builtin_bit_cast(int, v)));
}


// ---- LayerNorm(+ReLU) backward of four columns of one row (device) ------------------------------------------------------------
// Shared by ln_relu_bwd_vec_kernel (sage_linear.hip) and the LayerNorm-backward epilogue of the planes NT GEMM (gemm_p3.hip): the
// same instruction sequence at both call sites (explicit fmaf, contraction off for everything else), so the fused launch is bit
// for bit the two launches.  pre: xhat, masked gradient g and this lane's share of the two row sums; post (after the row sums
// c1 = mean(dxhat), c2 = mean(dxhat xhat) are known): dz and the column partials.
__device__ __forceinline__ void gte_ln_bwd_pre4(const float (&gy)[4], const float (&zz)[4], float mean, float rstd, const float (&gam)[4],
                                                const float (&bet)[4], bool okc, int relu, float (&xh)[4], float (&g)[4], float& a,
                                                float& b) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = okc ? (zz[e] - mean) * rstd : 0.f;
        float gv = gy[e];
        if (relu && fmaf(xh[e], gam[e], bet[e]) <= 0.f) gv = 0.f;
        g[e] = gv;
        const float dxh = gv * gam[e];
        a = a + dxh;
        b = fmaf(dxh, xh[e], b);
    }
}
__device__ __forceinline__ void gte_ln_bwd_post4(const float (&g)[4], const float (&xh)[4], const float (&gam)[4], float rstd, float c1,
                                                 float c2, bool okc, float (&d)[4], float (&s_dg)[4], float (&s_db)[4],
                                                 float (&s_dbias)[4]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t0 = g[e] * gam[e];
        const float t1 = xh[e] * c2;
        d[e] = rstd * ((t0 - c1) - t1);
        s_dg[e] = fmaf(g[e], xh[e], s_dg[e]);
        s_db[e] = s_db[e] + g[e];
        s_dbias[e] = s_dbias[e] + (okc ? d[e] : 0.f);
    }
}
// ... for a LayerNorm width that is not a multiple of 4 (16): per-element validity; the arithmetic of ln_relu_bwd_gen_kernel
__device__ __forceinline__ void gte_ln_bwd_pre4m(const float (&gy)[4], const float (&zz)[4], float mean, float rstd, const float (&gam)[4],
                                                 const float (&bet)[4], const bool (&ok)[4], int relu, float (&xh)[4], float (&g)[4],
                                                 float& a, float& b) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = ok[e] ? (zz[e] - mean) * rstd : 0.f;
        float gv = ok[e] ? gy[e] : 0.f;
        if (relu && fmaf(xh[e], gam[e], bet[e]) <= 0.f) gv = 0.f;
        g[e] = gv;
        const float dxh = gv * gam[e];
        a = a + dxh;
        b = fmaf(dxh, xh[e], b);
    }
}
__device__ __forceinline__ void gte_ln_bwd_post4m(const float (&g)[4], const float (&xh)[4], const float (&gam)[4], float rstd, float c1,
                                                  float c2, const bool (&ok)[4], float (&d)[4], float (&s_dg)[4], float (&s_db)[4],
                                                  float (&s_dbias)[4]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t0 = g[e] * gam[e];
        const float t1 = xh[e] * c2;
        d[e] = ok[e] ? rstd * ((t0 - c1) - t1) : 0.f;
        s_dg[e] = fmaf(g[e], xh[e], s_dg[e]);
        s_db[e] = s_db[e] + g[e];
        s_dbias[e] = s_dbias[e] + d[e];
    }
}
