// Shared pieces of the weighted cross-entropy (reference nn.CrossEntropyLoss(weight), src/models/model_train.py:171,327):
// the fixed-order fold of per-block partials {sum w*nll, sum w, #correct} that every consumer repeats for itself.
#pragma once
#include "gte_common.h"

namespace gte_ce {

constexpr int kCeBlock = 256;

template <typename L>
__device__ __forceinline__ int label_of(const L* labels, int64_t i) { return (int)labels[i]; }

// every thread of a kCeBlock-thread block takes part; result in red[.][0]
__device__ __forceinline__ void ce_fold(const float* __restrict__ partial, int64_t nblocks, double (&red)[3][kCeBlock]) {
    double a = 0., b = 0., d = 0.;
    for (int64_t i = threadIdx.x; i < nblocks; i += kCeBlock) {
        a += partial[i * 3 + 0];
        b += partial[i * 3 + 1];
        d += partial[i * 3 + 2];
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = d;
    __syncthreads();
    for (int s = kCeBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
            red[2][threadIdx.x] += red[2][threadIdx.x + s];
        }
        __syncthreads();
    }
}

// out3 = {loss = sum w*nll / sum w, sum w, #correct}
__device__ __forceinline__ void ce_write_out3(const double (&red)[3][kCeBlock], float* __restrict__ out3) {
    out3[0] = (float)(red[1][0] > 0. ? red[0][0] / red[1][0] : 0.);
    out3[1] = (float)red[1][0];
    out3[2] = (float)red[2][0];
}

}  // namespace gte_ce
