"""Yardstick only (never on the product path): the vendor fp32 GEMM (rocBLAS / hipBLASLt behind torch.matmul) at
the page-batch shapes, next to libgte_hip's kernels.  usage: python profiles/vendor_gemm_yardstick.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_tableextraction_amd import ops

torch.backends.cuda.matmul.allow_tf32 = False
dev, M = "cuda:0", 24495
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def row(name, flops, ours, vendor):
    print(f"{name:44s} gte {ours*1e3:8.1f} us {flops/ours/1e9:6.1f} TF | vendor {vendor*1e3:8.1f} us {flops/vendor/1e9:6.1f} TF", flush=True)


for pref in ("default", "hipblaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(pref)
    except Exception as ex:
        print("preferred_blas_library", pref, "->", ex); continue
    print("== vendor backend:", pref)
    for (k, n) in [(1662, 256), (512, 256), (2000, 1000)]:
        a, w = torch.randn(M, k, device=dev), torch.randn(n, k, device=dev) * 0.02
        o = torch.empty(M, n, device=dev)
        ours = timeit(lambda: ops.gemm(a, w, trans_b=True, out=o))
        ven = timeit(lambda: torch.matmul(a, w.t(), out=o))
        row(f"NT  M={M} K={k} N={n}", 2.0 * M * k * n, ours, ven)
    for (mo, n) in [(256, 1662), (256, 512), (1000, 1000)]:
        dz, x = torch.randn(M, mo, device=dev), torch.randn(M, n, device=dev)
        o = torch.empty(mo, n, device=dev)
        ours = timeit(lambda: ops.gemm(dz, x, trans_a=True, out=o))
        ven = timeit(lambda: torch.matmul(dz.t(), x, out=o))
        row(f"TN  M={mo} N={n} K={M}", 2.0 * M * mo * n, ours, ven)
    for (k, n) in [(256, 512), (256, 256), (1000, 1000)]:
        dz, w = torch.randn(M, k, device=dev), torch.randn(k, n, device=dev)
        o = torch.empty(M, n, device=dev)
        ours = timeit(lambda: ops.gemm(dz, w, out=o))
        ven = timeit(lambda: torch.matmul(dz, w, out=o))
        row(f"NN  M={M} N={n} K={k}", 2.0 * M * k * n, ours, ven)
