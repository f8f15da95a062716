// libgte_hip.so: error reporting, version and device facts.
#include "gte_common.h"
#include "batch_assemble.h"
#include "fold_images.h"
#include "p3.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

namespace gte {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

const DeviceProps& device_props() {
    // read-only cache, filled once (C++11 magic static: thread-safe)
    static const DeviceProps props = [] {
        DeviceProps p;
        p.cus = 256;
        p.lds_bytes = 160 * 1024;
        strncpy(p.arch, "unknown", sizeof(p.arch));
        int dev = 0;
        hipDeviceProp_t hp;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&hp, dev) == hipSuccess) {
            p.cus = hp.multiProcessorCount;
            p.lds_bytes = (int)hp.maxSharedMemoryPerMultiProcessor;
            strncpy(p.arch, hp.gcnArchName, sizeof(p.arch) - 1);
            p.arch[sizeof(p.arch) - 1] = 0;
        }
        return p;
    }();
    return props;
}

// ---- deferred folds ------------------------------------------------------------------------------------------
// A training step ends five kernels with "sum the per-block partials": LayerNorm column sums (x2), the narrow layer's
// dW, two split-K dW GEMMs.  Each fold is a few hundred KB to a few MB and latency-bound (6-11 us as its own launch,
// plus the launch gap); queued and run together they overlap each other: one launch before the optimizer.
constexpr int kMaxFolds = 24;
struct FoldDesc {
    const float* src; long long stride; float* dst; long long ld;
    int count, rows, rowlen, first_block, slices, vec;
    unsigned img_mask;                    // weight images (FoldImages) whose source overlaps this fold's destination
};
// Optional optimiser tail of the batch (gte_fold_defer_flush_adam): when `param` is set every folded element is a gradient
// element of the flat buffer starting at `grad`, and the thread that writes it applies the Adam update of that element.
struct FoldAdam { float* param; float* grad; float* exp_avg; float* exp_avg_sq; float* state; long long* step; unsigned* ticket; int vec_ok; };
struct FoldBatch { FoldDesc d[kMaxFolds]; int n; FoldAdam adam; FoldImages img; int fold_blocks; };
struct FoldQueue {
    FoldBatch batch; bool open = false; hipStream_t stream = nullptr; int blocks = 0;
    // the next batch's assembly, carried by the flush launch as workgroups behind the folds' (batch_assemble.h: defer_assemble)
    gte_asm::AssembleJob asm_job = {}; bool has_asm = false;
    bool spilled = false;                 // a full batch was flushed early: the queued folds no longer cover the whole deferral
    // coverage cache of gte_fold_defer_flush_adam: the descriptor signature it was computed for and its verdict
    unsigned long long cover_key = 0; bool cover_ok = false;
};
static FoldQueue& fold_queue() {
    static thread_local FoldQueue q;
    return q;
}

}  // namespace gte

// block = 256 threads = `slices` x (256 / slices) elements.  slice s sums partials s, s + slices, ... (independent loads
// in flight), the slice sums are added in slice order through LDS.  A split-K slab set (tens of partials, ~0.5 M
// elements) runs with one slice -- a thread streams its element through all partials; a block-partial set (hundreds
// of partials, a few thousand elements) with 16.
__global__ void __launch_bounds__(256)
gte_fold_batch_kernel(const gte::FoldBatch fb, const gte_asm::AssembleJob aj) {
    __shared__ __attribute__((aligned(16))) float part[4 * 256];
    if ((int)blockIdx.x >= fb.fold_blocks) {
        // a workgroup of the co-launched batch assembly (independent of the folds: another buffer set, read by the NEXT step)
        gte_asm::assemble_block(aj.a, aj.feat_wgs, aj.rows_per_wg, aj.n_out, (int)blockIdx.x - fb.fold_blocks);
        if (fb.adam.param) gte::adam_advance(fb.adam.state, fb.adam.step, fb.adam.ticket);      // (every workgroup of the grid checks in)
        return;
    }
    int di = 0;
#pragma unroll 1
    for (int i = 1; i < fb.n; ++i) if ((int)blockIdx.x >= fb.d[i].first_block) di = i;
    const gte::FoldDesc d = fb.d[di];
    const gte::FoldAdam ad = fb.adam;
    gte::AdamCoef co;
    if (ad.param) co = gte::adam_coef(ad.state);
    auto put = [&](float* dst, float v) {
        *dst = v;
        if (ad.param) {
            const long long i = dst - ad.grad;
            gte::adam_update(co, ad.param[i], v, ad.exp_avg[i], ad.exp_avg_sq[i]);
            if (d.img_mask) gte::fold_write_images(fb.img, d.img_mask, i, ad.param[i]);
        }
    };
    const int epb = 256 / d.slices;
    const int el = threadIdx.x % epb, sl = threadIdx.x / epb;
    const long long total = (long long)d.rows * d.rowlen;
    if (d.vec == 4) {
        // four consecutive elements per thread, 16-byte loads; per element the same partial order as the scalar path
        const long long e = ((long long)((int)blockIdx.x - d.first_block) * epb + el) * 4;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0, s6 = s0, s7 = s0;
        auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
        // contiguous, 16-byte aligned destination with the optimiser tail: the four elements' parameter and moments move as
        // 16-byte accesses, requested BEFORE the partials are streamed (one memory round trip less on the critical path)
        float* d0 = d.dst + e;
        const bool writer = sl == 0 && e < total;
        const bool fast = writer && ad.param && ad.vec_ok && d.ld == d.rowlen && (reinterpret_cast<uintptr_t>(d0) & 15) == 0;
        const long long ai = fast ? d0 - ad.grad : 0;
        float4 pv = s0, mv = s0, qv = s0;
        if (fast) {
            pv = *reinterpret_cast<const float4*>(ad.param + ai);
            mv = *reinterpret_cast<const float4*>(ad.exp_avg + ai);
            qv = *reinterpret_cast<const float4*>(ad.exp_avg_sq + ai);
        }
        if (e < total) {
            const float* p = d.src + e;
            const long long st = d.stride * d.slices;
            int k = sl;
            // eight partials in flight per thread: the launch lasts as long as its longest chain of dependent memory round
            // trips (r03: 512 LayerNorm block partials in 16 slices were 8 rounds of four loads, ~2 us each)
            for (; k + 7 * d.slices < d.count; k += 8 * d.slices) {
                const float* pk = p + k * d.stride;
                const float4 v0 = *reinterpret_cast<const float4*>(pk), v1 = *reinterpret_cast<const float4*>(pk + st),
                             v2 = *reinterpret_cast<const float4*>(pk + 2 * st), v3 = *reinterpret_cast<const float4*>(pk + 3 * st),
                             v4 = *reinterpret_cast<const float4*>(pk + 4 * st), v5 = *reinterpret_cast<const float4*>(pk + 5 * st),
                             v6 = *reinterpret_cast<const float4*>(pk + 6 * st), v7 = *reinterpret_cast<const float4*>(pk + 7 * st);
                add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3); add(s4, v4); add(s5, v5); add(s6, v6); add(s7, v7);
            }
            for (; k + 3 * d.slices < d.count; k += 4 * d.slices) {
                const float* pk = p + k * d.stride;
                add(s0, *reinterpret_cast<const float4*>(pk));
                add(s1, *reinterpret_cast<const float4*>(pk + st));
                add(s2, *reinterpret_cast<const float4*>(pk + 2 * st));
                add(s3, *reinterpret_cast<const float4*>(pk + 3 * st));
            }
            for (; k < d.count; k += d.slices) add(s0, *reinterpret_cast<const float4*>(p + k * d.stride));
        }
        float4 t;
        add(s0, s4); add(s1, s5); add(s2, s6); add(s3, s7);
        t.x = (s0.x + s1.x) + (s2.x + s3.x); t.y = (s0.y + s1.y) + (s2.y + s3.y);
        t.z = (s0.z + s1.z) + (s2.z + s3.z); t.w = (s0.w + s1.w) + (s2.w + s3.w);
        reinterpret_cast<float4*>(part)[threadIdx.x] = t;
        __syncthreads();
        if (writer) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int u = 0; u < d.slices; ++u) add(v, reinterpret_cast<const float4*>(part)[u * epb + el]);
            if (fast) {
                *reinterpret_cast<float4*>(d0) = v;
                gte::adam_update(co, pv.x, v.x, mv.x, qv.x); gte::adam_update(co, pv.y, v.y, mv.y, qv.y);
                gte::adam_update(co, pv.z, v.z, mv.z, qv.z); gte::adam_update(co, pv.w, v.w, mv.w, qv.w);
                *reinterpret_cast<float4*>(ad.exp_avg + ai) = mv;
                *reinterpret_cast<float4*>(ad.exp_avg_sq + ai) = qv;
                *reinterpret_cast<float4*>(ad.param + ai) = pv;
                if (d.img_mask) {
                    // (pairs: one 8-byte store per plane for the four measured slower -- rows of odd length leave them 4-byte aligned)
                    gte::fold_write_images2(fb.img, d.img_mask, ai, pv.x, pv.y);
                    gte::fold_write_images2(fb.img, d.img_mask, ai + 2, pv.z, pv.w);
                }
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long ee = e + j, r = ee / d.rowlen;
                    put(&d.dst[r * d.ld + (ee - r * d.rowlen)], vv[j]);
                }
            }
        }
        if (ad.param) gte::adam_advance(ad.state, ad.step, ad.ticket);
        return;
    }
    const long long e = (long long)((int)blockIdx.x - d.first_block) * epb + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < total) {
        const float* p = d.src + e;
        const long long st = d.stride * d.slices;
        int k = sl;
        float s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
        for (; k + 7 * d.slices < d.count; k += 8 * d.slices) {
            const float* pk = p + k * d.stride;
            const float v0 = pk[0], v1 = pk[st], v2 = pk[2 * st], v3 = pk[3 * st], v4 = pk[4 * st], v5 = pk[5 * st], v6 = pk[6 * st],
                        v7 = pk[7 * st];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
        }
        s0 += s4; s1 += s5; s2 += s6; s3 += s7;
        for (; k + 3 * d.slices < d.count; k += 4 * d.slices) {
            const float* pk = p + k * d.stride;
            s0 += pk[0]; s1 += pk[st]; s2 += pk[2 * st]; s3 += pk[3 * st];
        }
        for (; k < d.count; k += d.slices) s0 += p[k * d.stride];
    }
    part[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && e < total) {
        float v = 0.f;
        for (int t = 0; t < d.slices; ++t) v += part[t * epb + el];
        const long long r = e / d.rowlen;
        put(&d.dst[r * d.ld + (e - r * d.rowlen)], v);
    }
    if (ad.param) gte::adam_advance(ad.state, ad.step, ad.ticket);
}

namespace gte {

static int flush_folds(FoldQueue& q, const FoldAdam* adam = nullptr, const FoldImages* images = nullptr) {
    if (q.batch.n == 0 && !q.has_asm) return GTE_OK;
    if (q.batch.n == 0) adam = nullptr;                 // (a deferred assembly alone: no folds, no optimiser tail)
    q.batch.adam = adam ? *adam : FoldAdam{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (images) q.batch.img = *images; else q.batch.img.n = 0;
    for (int i = 0; i < q.batch.n; ++i) {
        FoldDesc& d = q.batch.d[i];
        d.img_mask = 0;
        if (!images || !adam) continue;
        const long long lo = d.dst - adam->grad, hi = lo + (long long)(d.rows - 1) * d.ld + d.rowlen;
        for (int k = 0; k < images->n; ++k)
            if (images->im[k].off < hi && images->im[k].off + (long long)images->im[k].span > lo) d.img_mask |= 1u << k;
    }
    q.batch.fold_blocks = q.blocks;
    const gte_asm::AssembleJob none = {};
    const int asm_blocks = q.has_asm ? q.asm_job.blocks : 0;
    hipLaunchKernelGGL(gte_fold_batch_kernel, dim3((unsigned)(q.blocks + asm_blocks)), dim3(256), 0, q.stream, q.batch,
                       q.has_asm ? q.asm_job : none);
    q.batch.n = 0;
    q.blocks = 0;
    q.has_asm = false;
    return check_launch("fold_batch");
}

bool defer_assemble(const gte_asm::AssembleJob& job, hipStream_t stream) {
    FoldQueue& q = fold_queue();
    if (!q.open || q.stream != stream || q.has_asm || job.blocks <= 0) return false;
    q.asm_job = job;
    q.has_asm = true;
    return true;
}

bool defer_fold(const float* src, int64_t stride, int count, int rows, int rowlen, float* dst, int64_t ld) {
    FoldQueue& q = fold_queue();
    if (!q.open) return false;
    if (!dst || rows <= 0 || rowlen <= 0) return true;
    if (q.batch.n == kMaxFolds) { (void)flush_folds(q); q.spilled = true; }
    FoldDesc& d = q.batch.d[q.batch.n++];
    d.src = src; d.stride = stride; d.dst = dst; d.ld = ld;
    d.count = count; d.rows = rows; d.rowlen = rowlen; d.first_block = q.blocks;
    // (slices for 12..31 partials -- the split-K slabs -- measured: 4 slices 42.9 us against 27.8 us with one: the optimiser
    // tail then runs in a quarter of the threads)
    d.slices = count >= 128 ? 16 : (count >= 32 ? 4 : 1);
    const int64_t total = (int64_t)rows * rowlen;
    d.vec = (total % 4 == 0 && stride % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && total >= 4096) ? 4 : 1;
    q.blocks += (int)ceil_div(total / d.vec, 256 / d.slices);
    return true;
}

}  // namespace gte

namespace gte {
static TailWorkspace& tail_ws_slot() {
    static thread_local TailWorkspace w = {nullptr, 0};
    return w;
}
TailWorkspace tail_workspace() { return tail_ws_slot(); }
}  // namespace gte

extern "C" int64_t gte_gemm_tail_workspace_bytes(void) {
    // at most one partial tile (128 x 128 fp32) per compute unit
    return (int64_t)gte::device_props().cus * 128 * 128 * 4;
}

extern "C" int gte_gemm_set_tail_workspace(void* workspace, int64_t workspace_bytes) {
    if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace))
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "gemm_set_tail_workspace: bad arguments");
    gte::tail_ws_slot() = {reinterpret_cast<float*>(workspace), workspace ? workspace_bytes : 0};
    return GTE_OK;
}

extern "C" int gte_fold_defer_begin(void* stream) {
    gte::FoldQueue& q = gte::fold_queue();
    if (q.open) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "fold_defer_begin: a deferral is already open on this thread");
    q.open = true;
    q.stream = gte::as_stream(stream);
    q.batch.n = 0;
    q.blocks = 0;
    return GTE_OK;
}

extern "C" int gte_fold_defer_flush(void) {
    gte::FoldQueue& q = gte::fold_queue();
    if (!q.open) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "fold_defer_flush: no deferral is open on this thread");
    q.open = false;
    q.spilled = false;
    return gte::flush_folds(q);
}

namespace gte {
// Do the queued folds write every element of [grad, grad + n) exactly once?  Rows of a fold are intervals
// [off + r ld, off + r ld + rowlen); sorted and walked once.  The answer is cached on a signature of the descriptors (the
// same step queues the same folds), so the steady state costs one pass over <= 24 descriptors.
static bool folds_cover(FoldQueue& q, const float* grad, int64_t n) {
    unsigned long long key = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
    mix((unsigned long long)(uintptr_t)grad); mix((unsigned long long)n); mix((unsigned long long)q.batch.n);
    for (int i = 0; i < q.batch.n; ++i) {
        const FoldDesc& d = q.batch.d[i];
        mix((unsigned long long)(uintptr_t)d.dst); mix((unsigned long long)d.ld);
        mix(((unsigned long long)d.rows << 32) | (unsigned)d.rowlen);
    }
    if (key == q.cover_key) return q.cover_ok;
    std::vector<std::pair<int64_t, int64_t>> iv;
    bool ok = true;
    for (int i = 0; i < q.batch.n && ok; ++i) {
        const FoldDesc& d = q.batch.d[i];
        const int64_t off = d.dst - grad;
        if (d.dst < grad || off + (int64_t)(d.rows - 1) * d.ld + d.rowlen > n) { ok = false; break; }
        if (d.ld == d.rowlen) iv.emplace_back(off, off + (int64_t)d.rows * d.rowlen);
        else for (int r = 0; r < d.rows; ++r) iv.emplace_back(off + (int64_t)r * d.ld, off + (int64_t)r * d.ld + d.rowlen);
    }
    if (ok) {
        std::sort(iv.begin(), iv.end());
        int64_t at = 0;
        for (const auto& v : iv) {
            if (v.first != at) { ok = false; break; }      // a gap (an element no fold writes) or an overlap
            at = v.second;
        }
        ok = ok && at == n;
    }
    q.cover_key = key;
    q.cover_ok = ok;
    return ok;
}
}  // namespace gte

extern "C" int gte_fold_defer_flush_adam_images(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                                int64_t* step_counter, unsigned* ticket, const gte_p3_desc* images, int n_images,
                                                int* fused) {
    gte::FoldQueue& q = gte::fold_queue();
    if (fused) *fused = 0;
    if (!q.open) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "fold_defer_flush_adam: no deferral is open on this thread");
    q.open = false;
    const bool spilled = q.spilled;
    q.spilled = false;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || !step_counter || !ticket || !fused || n <= 0) {
        (void)gte::flush_folds(q);
        return gte::fail(GTE_ERR_INVALID_ARGUMENT, "fold_defer_flush_adam: null pointer or n <= 0");
    }
    gte::FoldImages fi;
    {
        const int rc_img = gte::make_fold_images(param, n, images, n_images, fi, "fold_defer_flush_adam_images");
        if (rc_img != GTE_OK) { (void)gte::flush_folds(q); return rc_img; }
    }
    if (spilled || q.batch.n == 0 || !gte::folds_cover(q, grad, n)) return gte::flush_folds(q);   // caller runs gte_adam_step_dev
    const int vec_ok = ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
    const gte::FoldAdam ad = {param, grad, exp_avg, exp_avg_sq, state, reinterpret_cast<long long*>(step_counter), ticket, vec_ok};
    const int rc = gte::flush_folds(q, &ad, fi.n > 0 ? &fi : nullptr);
    if (rc == GTE_OK) *fused = fi.n > 0 ? 3 : 1;
    return rc;
}

extern "C" int gte_fold_defer_flush_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                         int64_t* step_counter, unsigned* ticket, int* fused) {
    return gte_fold_defer_flush_adam_images(param, grad, exp_avg, exp_avg_sq, n, state, step_counter, ticket, nullptr, 0, fused);
}

extern "C" int gte_version(void) { return GTE_VERSION; }

extern "C" const char* gte_last_error(void) { return gte::error_buffer(); }

extern "C" int gte_device_info(int* compute_units, int* wave_size, int* lds_bytes, char* arch_name,
                               int arch_name_len) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return gte::fail(GTE_ERR_UNSUPPORTED, "gte_device_info: no HIP device visible");
    const gte::DeviceProps& p = gte::device_props();
    if (compute_units) *compute_units = p.cus;
    if (wave_size) *wave_size = gte::kWave;
    if (lds_bytes) *lds_bytes = p.lds_bytes;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.arch, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return GTE_OK;
}
