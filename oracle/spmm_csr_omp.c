/* CPU oracle: row-parallel CSR SpMM  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Restates the aggregation the reference reaches through DGL at
 * src/components/graphs/models.py:53-54
 *     g.update_all(fn.u_mul_e('h','feat','m'), fn.sum(msg='m', out='h'))
 * i.e. out[v,:] = sum_{e=(u->v)} w_e * x[u,:]  (0 for in-degree-0 rows), and the
 * mean reducer of models.py:146-149 (mode 1: divide by the in-degree).
 * DGL is a third-party dependency that is not vendored (no pinned version); this
 * follows its published CPU algorithm (SpMMSumCsr: OpenMP over destination rows
 * of the in-edge CSR, sequential accumulation along each row).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 * Parity: pinned against the stub-dgl golden vectors in tests/golden and the
 * fp64 dense formulation in oracle/gcnsage_cpu.py; the DGL kernel itself could
 * not be run here ("parity unpinned" for that operator beyond its semantics).
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* mode 0: weighted sum; mode 1: weighted sum / in_degree (0 if none). */
void oracle_spmm_csr_f32(const int32_t* indptr, const int32_t* indices, const float* w,
                         const float* x, float* out, int64_t n, int64_t f,
                         int64_t ldx, int64_t ldo, int mode) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t v = 0; v < n; ++v) {
        float* o = out + v * ldo;
        memset(o, 0, (size_t)f * sizeof(float));
        const int32_t lo = indptr[v], hi = indptr[v + 1];
        for (int32_t e = lo; e < hi; ++e) {
            const float* xr = x + (int64_t)indices[e] * ldx;
            const float we = w ? w[e] : 1.0f;
            for (int64_t j = 0; j < f; ++j) o[j] += we * xr[j];
        }
        if (mode == 1 && hi > lo) {
            const float inv = 1.0f / (float)(hi - lo);
            for (int64_t j = 0; j < f; ++j) o[j] *= inv;
        }
    }
}
