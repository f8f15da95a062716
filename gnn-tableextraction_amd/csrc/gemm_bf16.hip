// bf16 projection GEMM for the GAT of BASELINE configs[2] ("4-head GAT bf16"; SURVEY A13 -- no reference symbol, parity
// unpinned): z = X W^T with bf16 operands on v_mfma_f32_32x32x16_bf16 and fp32 accumulation, plus the fp32 -> bf16 casts
// (optionally through ELU) that feed it.
//
// Roofline: at the widths of this model (K, N <= 1024 against ~10^5 rows) the projection is HBM-bound even on the bf16 matrix
// pipe -- 2 K N flop against (2 K + 4 N) bytes per row is 50-170 flop/B, the bf16 ridge is ~310 flop/B -- so the kernel is
// built for coalesced 16-byte streams, not for the last percent of the matrix pipe: 128 x 128 block tile, 4 waves of 2 x 2
// MFMA tiles, BK = 32 (two K = 16 MFMA steps), operands K-contiguous in LDS with an 8-element pad (conflict-free ds_read_b128:
// a lane's 8 consecutive k of one row ARE the MFMA operand), next stage's global loads in flight under the current stage's
// MFMAs.  Operands must be 16-byte aligned with leading dimensions and K multiples of 8 (the cast pads with zeros).
#include "gte_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

constexpr int BM = 128, BN = 128, BKB = 32, LDP = BKB + 8;      // LDS row stride in bf16 (80 bytes: 16-byte aligned, conflict-free)

__device__ __forceinline__ u16 f2bf(float x) {
    __bf16 b = (__bf16)x;                                        // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return *reinterpret_cast<u16*>(&b);
}

// y[r, 0:cols_pad] = bf16(act(x[r, 0:cols])), zero padded to cols_pad (multiple of 8).  act: 0 none, 1 ELU(alpha = 1)
__global__ void __launch_bounds__(256)
cast_bf16_kernel(const float* __restrict__ x, int64_t ldx, u16* __restrict__ y, int64_t ldy, int64_t rows, int cols, int cols_pad,
                 int act) {
    const int64_t groups = cols_pad / 8;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * groups) return;
    const int64_t r = i / groups;
    const int c0 = (int)(i % groups) * 8;
    u16 o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = (c0 + j < cols) ? x[r * ldx + c0 + j] : 0.f;
        if (act == 1) v = v > 0.f ? v : expm1f(v);
        o[j] = f2bf(v);
    }
    *reinterpret_cast<uint4*>(y + r * ldy + c0) = *reinterpret_cast<const uint4*>(o);
}

// C[M, N] (fp32) = A[M, K] B[N, K]^T, bf16 operands.  grid = tiles, XCD-aware (tiles sharing an A row panel on one XCD).
__global__ void __launch_bounds__(256)
gemm_bf16_nt_kernel(const u16* __restrict__ A, int64_t lda, const u16* __restrict__ B, int64_t ldb, float* __restrict__ C,
                    int64_t ldc, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) u16 sA[BM * LDP];
    __shared__ __attribute__((aligned(16))) u16 sB[BN * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const unsigned lb = gte_xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n));
    const int m0 = (int)(lb / tiles_n) * BM, n0 = (int)(lb % tiles_n) * BN;

    // staging: a tile is 128 rows x 32 bf16 = 512 16-byte pieces; thread t owns pieces t and t + 256 (rows t/4 and t/4 + 64)
    const int pr = tid >> 2, pc = (tid & 3) * 8;
    uint4 ra[2], rb[2];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ar = m0 + pr + 64 * h, br = n0 + pr + 64 * h;
            ra[h] = (ar < M && k0 + pc < K) ? *reinterpret_cast<const uint4*>(A + (int64_t)ar * lda + k0 + pc) : make_uint4(0, 0, 0, 0);
            rb[h] = (br < N && k0 + pc < K) ? *reinterpret_cast<const uint4*>(B + (int64_t)br * ldb + k0 + pc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<uint4*>(&sA[(pr + 64 * h) * LDP + pc]) = ra[h];
            *reinterpret_cast<uint4*>(&sB[(pr + 64 * h) * LDP + pc]) = rb[h];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_stage(0);
    store_stage();
    __syncthreads();
    const int i32 = lane & 31, kh = (lane >> 5) * 8;
    for (int k0 = 0; k0 < K; k0 += BKB) {
        if (k0 + BKB < K) load_stage(k0 + BKB);                  // in flight under this stage's MFMAs
#pragma unroll
        for (int ks = 0; ks < BKB; ks += 16) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(&sA[((wm * 2 + a) * 32 + i32) * LDP + ks + kh]);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(&sB[((wn * 2 + b) * 32 + i32) * LDP + ks + kh]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (k0 + BKB < K) store_stage();
        __syncthreads();
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int hrow = (lane >> 5) * 4;
    // branch-free stores through a buffer descriptor over C (rows past M / columns past N fall outside the window and are
    // dropped): see the store epilogue of sage_linear.hip
    if (((int64_t)M + BM) * ldc * 4 < ((int64_t)1 << 31)) {
        const __amdgpu_buffer_rsrc_t c_srd = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)(((int64_t)(M - 1) * ldc + N) * 4), 0x00020000u);
        const int ld4 = (int)ldc * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i32;
            const int coff = col < N ? col * 4 : (int)0x80000000;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int roff0 = (m0 + (wm * 2 + a) * 32 + hrow) * ld4 + coff;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // through a VGPR: with the accumulator element (an AGPR) as the store's data operand hipcc 7.2 emitted
                    // the SAME register (element 0) for all sixteen stores
                    float v = acc[a][b][r];
                    asm volatile("" : "+v"(v));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_srd,
                                                          roff0 + ((r & 3) + 8 * (r >> 2)) * ld4, 0, 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i32;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + a) * 32 + hrow + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(int64_t)row * ldc + col] = acc[a][b][r];
            }
        }
}

}  // namespace

extern "C" int gte_cast_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int64_t rows, int64_t cols, int activation,
                             void* stream) {
    if (rows < 0 || cols <= 0 || cols > INT32_MAX) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "cast_bf16: bad sizes");
    if (rows == 0) return GTE_OK;
    if (!x || !y) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "cast_bf16: null pointer");
    const int64_t cols_pad = gte::round_up(cols, 8);
    if (ldx < cols || ldy < cols_pad || ldy % 8 != 0 || ((uintptr_t)y & 15) != 0)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "cast_bf16: y must be 16-byte aligned with ldy a multiple of 8 >= cols rounded up to 8");
    if (activation != 0 && activation != 1) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "cast_bf16: activation must be 0 (none) or 1 (ELU)");
    const int64_t work = rows * (cols_pad / 8);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)gte::ceil_div(work, 256)), dim3(256), 0, gte::as_stream(stream), x, ldx, y,
                       ldy, rows, (int)cols, (int)cols_pad, activation);
    return gte::check_launch("cast_bf16");
}

extern "C" int gte_gemm_bf16_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                                int64_t N, int64_t K, void* stream) {
    if (M < 0 || N < 0 || K < 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_bf16_nt: bad sizes");
    if (M == 0 || N == 0) return GTE_OK;
    if (!A || !B || !C) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_bf16_nt: null pointer");
    if (K % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 || lda < K || ldb < K || ldc < N || ((uintptr_t)A & 15) != 0 || ((uintptr_t)B & 15) != 0)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_bf16_nt: operands must be 16-byte aligned, K and leading dimensions multiples of 8");
    const int64_t tiles = gte::ceil_div(M, BM) * gte::ceil_div(N, BN);
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3((unsigned)tiles), dim3(256), 0, gte::as_stream(stream), A, lda, B, ldb, C, ldc, (int)M,
                       (int)N, (int)K);
    return gte::check_launch("gemm_bf16_nt");
}
