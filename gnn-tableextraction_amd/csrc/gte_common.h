// Shared host-side helpers for libgte_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

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
