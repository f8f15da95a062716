// P3: an fp32 matrix stored as three bf16 planes, the operand format of the planes GEMMs (gemm_p3.hip).
//
// An fp32 number x is EXACTLY h + m + l with three bf16 numbers (3 x 8 significand bits):
//     h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (round to nearest even; the subtractions are exact in fp32)
// (csrc/gemm_split.h makes this cut inside the GEMM, once per operand tile and workgroup; here the PRODUCER of an operand
// makes it once, in the epilogue that writes the operand, and the GEMM moves bf16 planes straight from memory to LDS.)
//
// Layout of a logical [R][F] matrix, Fb = ceil(F / 16) blocks of 16 features per row:
//     row r at byte r * ldp (ldp >= 96 Fb, a multiple of 16);  block fb at + 96 fb;
//     inside a block: plane h = 16 bf16 (32 bytes), plane m (32 bytes), plane l (32 bytes);  features >= F are zero.
// One layout serves both operand roles of a matrix:
//   * K = feature index (forward transform, dX): a K stage of 16 is one block = 96 contiguous bytes per row;
//   * K = row index (dW: reduction over the nodes): a stage of 16 rows x 128 features is 16 runs of 768 contiguous bytes, read
//     from LDS through the transposing ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace p3 {

constexpr int BLOCK = 16;            // features per block
constexpr int BLOCK_BYTES = 96;      // 3 planes x 16 bf16
constexpr int PLANE_BYTES = 32;

__host__ __device__ inline int64_t blocks(int64_t cols) { return (cols + BLOCK - 1) / BLOCK; }
__host__ __device__ inline int64_t row_bytes(int64_t cols) { return blocks(cols) * BLOCK_BYTES; }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// two fp32 -> their packed bf16 pieces (low half = first element).  Same instruction sequence as gemm_split.h's split_pair:
// the planes GEMM on producer-made planes is bit-identical to the split GEMM on the fp32 operand.
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t{x0, x1}), bf16x2_t));
    const float r10 = x0 - __builtin_bit_cast(float, h << 16);
    const float r11 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t{r10, r11}), bf16x2_t));
    const float r20 = r10 - __builtin_bit_cast(float, m << 16);
    const float r21 = r11 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t{r20, r21}), bf16x2_t));
}

// four consecutive features (a quarter block, 8 bytes per plane) of one row -> the row's P3 image.
// col0 % 4 == 0.  `row` points at the row's first byte.
__device__ __forceinline__ void store4(char* __restrict__ row, int col0, float x0, float x1, float x2, float x3) {
    uint2 h, m, l;
    split2(x0, x1, h.x, m.x, l.x);
    split2(x2, x3, h.y, m.y, l.y);
    char* dst = row + (col0 >> 4) * BLOCK_BYTES + (col0 & 15) * 2;
    *reinterpret_cast<uint2*>(dst) = h;
    *reinterpret_cast<uint2*>(dst + PLANE_BYTES) = m;
    *reinterpret_cast<uint2*>(dst + 2 * PLANE_BYTES) = l;
}

// eight consecutive features (half a block, 16 bytes per plane).  col0 % 8 == 0.
__device__ __forceinline__ void store8(char* __restrict__ row, int col0, const float (&x)[8]) {
    uint4 h, m, l;
    split2(x[0], x[1], h.x, m.x, l.x);
    split2(x[2], x[3], h.y, m.y, l.y);
    split2(x[4], x[5], h.z, m.z, l.z);
    split2(x[6], x[7], h.w, m.w, l.w);
    char* dst = row + (col0 >> 4) * BLOCK_BYTES + (col0 & 15) * 2;
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + PLANE_BYTES) = m;
    *reinterpret_cast<uint4*>(dst + 2 * PLANE_BYTES) = l;
}

// element (row image, col) back to fp32: (h + m) + l, exact
__device__ __forceinline__ float load1(const char* __restrict__ row, int col) {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(row + (col >> 4) * BLOCK_BYTES + (col & 15) * 2);
    const float h = __builtin_bit_cast(float, (unsigned)p[0] << 16);
    const float m = __builtin_bit_cast(float, (unsigned)p[PLANE_BYTES / 2] << 16);
    const float l = __builtin_bit_cast(float, (unsigned)p[PLANE_BYTES] << 16);
    return (h + m) + l;
}

}  // namespace p3
