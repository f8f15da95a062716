// Graph preparation on the device: COO -> CSR (in-edge or out-edge), 1/in-degree.
//
// replaces what DGL does lazily before gSpMM on a freshly batched graph (COO->CSC build for
// the forward, reverse CSR for GSpMM.backward) -- reference call sites: dgl.graph(...) at
// src/components/graphs/builder.py:425, dgl.batch(...) at src/models/model_train.py:246,297,
// and get_norm at src/components/graphs/models.py:74-78.  SURVEY 8(f) N1.
//
// Deterministic by construction: a STABLE radix sort of (key, edge id) pairs, so inside a row
// the entries are in ascending original edge id and the aggregation's summation order is
// fixed.  No atomics.  The sort itself is rocPRIM's device radix sort (header-only, compiled
// into this library); everything around it is plain HBM-bound index work.
#include "gte_common.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

__global__ void iota_kernel(int32_t* __restrict__ v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

// indptr[v] = first position in the sorted key array whose key >= v  (v = 0..n; indptr[n] = E)
__global__ void lower_bound_kernel(const int32_t* __restrict__ sorted_key, int64_t n_edges,
                                   int32_t* __restrict__ indptr, int64_t n_nodes) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n_nodes) return;
    int64_t lo = 0, hi = n_edges;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted_key[mid] < (int32_t)v) lo = mid + 1; else hi = mid;
    }
    indptr[v] = (int32_t)lo;
}

__global__ void gather_edges_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ other,
                                    const float* __restrict__ ew, const float* __restrict__ row_scale,
                                    int32_t* __restrict__ indices, float* __restrict__ wout, int64_t n_edges) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_edges) return;
    const int32_t e = perm[i];
    const int32_t o = other[e];
    indices[i] = o;
    if (wout) {
        float w = ew ? ew[e] : 1.0f;
        if (row_scale) w *= row_scale[o];
        wout[i] = w;
    }
}

__global__ void inv_degree_kernel(const int32_t* __restrict__ indptr, float* __restrict__ inv, int64_t n) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int d = indptr[v + 1] - indptr[v];
    inv[v] = d > 0 ? 1.0f / (float)d : 0.0f;      // models.py:74-78: 1/deg, inf -> 0
}

int key_bits(int64_t n_nodes) {
    int b = 1;
    while (((int64_t)1 << b) < n_nodes) ++b;
    return b;
}

size_t sort_temp_bytes(int64_t n_nodes, int64_t n_edges) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                    (int32_t*)nullptr, (size_t)n_edges, 0, key_bits(n_nodes), (hipStream_t)0);
    return bytes;
}

}  // namespace

extern "C" int64_t gte_coo_to_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    if (n_edges <= 0) return 256;
    const int64_t e4 = gte::round_up(n_edges * 4, 256);
    return 2 * e4 + (int64_t)gte::round_up((int64_t)sort_temp_bytes(n_nodes, n_edges), 256) + 256;
}

extern "C" int gte_coo_to_csr(const int32_t* key, const int32_t* other, const float* eweight,
                              const float* row_scale, int64_t n_nodes, int64_t n_edges, int32_t* indptr,
                              int32_t* indices, int32_t* perm, float* wout, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    if (n_nodes < 0 || n_edges < 0 || n_nodes >= INT32_MAX || n_edges >= INT32_MAX)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "coo_to_csr: bad sizes");
    if (!indptr) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "coo_to_csr: indptr is NULL");
    hipStream_t s = gte::as_stream(stream);
    if (n_edges == 0) {
        if (hipMemsetAsync(indptr, 0, (size_t)(n_nodes + 1) * 4, s) != hipSuccess)
            return gte::fail(GTE_ERR_LAUNCH, "coo_to_csr: memset failed");
        return GTE_OK;
    }
    if (!key || !other || !indices || !perm || !workspace)
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "coo_to_csr: null pointer");
    if (workspace_bytes < gte_coo_to_csr_workspace_bytes(n_nodes, n_edges))
        return gte::fail(GTE_ERR_WORKSPACE_TOO_SMALL, "coo_to_csr: workspace %lld < %lld", (long long)workspace_bytes,
                         (long long)gte_coo_to_csr_workspace_bytes(n_nodes, n_edges));
    const int64_t e4 = gte::round_up(n_edges * 4, 256);
    char* ws = reinterpret_cast<char*>(workspace);
    int32_t* sorted_key = reinterpret_cast<int32_t*>(ws);
    int32_t* ids = reinterpret_cast<int32_t*>(ws + e4);
    void* temp = ws + 2 * e4;
    size_t temp_bytes = sort_temp_bytes(n_nodes, n_edges);

    const int T = 256;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)gte::ceil_div(n_edges, T)), dim3(T), 0, s, ids, n_edges);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, key, sorted_key, ids, perm, (size_t)n_edges, 0,
                                             key_bits(n_nodes), s);
    if (e != hipSuccess) return gte::fail(GTE_ERR_LAUNCH, "coo_to_csr: radix sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(lower_bound_kernel, dim3((unsigned)gte::ceil_div(n_nodes + 1, T)), dim3(T), 0, s, sorted_key,
                       n_edges, indptr, n_nodes);
    hipLaunchKernelGGL(gather_edges_kernel, dim3((unsigned)gte::ceil_div(n_edges, T)), dim3(T), 0, s, perm, other,
                       eweight, row_scale, indices, wout, n_edges);
    return gte::check_launch("coo_to_csr");
}

extern "C" int gte_inv_degree(const int32_t* indptr, float* inv_deg, int64_t n_nodes, void* stream) {
    if (n_nodes < 0) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "inv_degree: n < 0");
    if (n_nodes == 0) return GTE_OK;
    if (!indptr || !inv_deg) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "inv_degree: null pointer");
    hipLaunchKernelGGL(inv_degree_kernel, dim3((unsigned)gte::ceil_div(n_nodes, 256)), dim3(256), 0,
                       gte::as_stream(stream), indptr, inv_deg, n_nodes);
    return gte::check_launch("inv_degree");
}
