"""Aggregation kernels on cfg4 (1M nodes, deg 12, F=512) and cfg2-sized page batches: plain vs LDS-staged."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import ops
from gnn_tableextraction_amd.data import synthetic as S
dev = "cuda:0"

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

def run(name, src, dst, w, n, feats):
    indptr, indices, perm, wout = ops.coo_to_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, torch.from_numpy(w).to(dev))
    plan = ops.build_tile_plan(indptr, indices, n)
    for f in feats:
        x = torch.randn(n, f, device=dev); out = torch.empty_like(x)
        alg = 2.0 * n * f * 4 + 8.0 * len(src) + 4.0 * (n + 1)
        a = timeit(lambda: ops.spmm_csr(indptr, indices, wout, x, n, mean=True, out=out))
        b = timeit(lambda: ops.spmm_csr(indptr, indices, wout, x, n, mean=True, out=out, tiles=plan, force_tiled=True))
        print(f"{name} F={f:4d}: plain {a*1e3:8.1f} us {alg/a/1e6:7.0f} GB/s | tiled {b*1e3:8.1f} us {alg/b/1e6:7.0f} GB/s", flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, dst, w = S.make_knn_stress_graph(n, 12)
run(f"knn{n}", src, dst, w, n, [512, 256, 128, 64])
pages = S.make_pages(100, in_feats=13)
src, dst, w, feat, label, off = S.concat_pages(pages)
run("pages100", src, dst, w, int(off[-1]), [831, 256, 64])
