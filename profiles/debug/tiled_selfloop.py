"""The LDS-staged aggregation as a pure copy: a graph where every node's only in-edge is a self loop (32 distinct sources per
32-row tile, no read amplification) vs in-degree 2 / 4 neighbour graphs: separates the cost of the 128-byte-piece access
pattern + staging from the cost of the re-reads through L2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_tableextraction_amd import ops
dev = "cuda:0"
n, f = 1_000_000, 512
x = torch.randn(n, f, device=dev); out = torch.empty_like(x)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for name, offs in (("self loop", [0]), ("12 x self loop", [0] * 12), ("2 nbrs", [0, 1]), ("4 nbrs", [-2, -1, 1, 2]), ("12 nbrs within +-6", list(range(-6, 0)) + list(range(1, 7))),
                   ("12 nbrs within +-40", [-40, -33, -21, -13, -5, -1, 1, 6, 14, 22, 31, 40])):
    dst = np.repeat(np.arange(n, dtype=np.int64), len(offs))
    src = (dst + np.tile(np.array(offs), n)).clip(0, n - 1)
    w = np.ones(len(src), np.float32)
    indptr, indices, perm, wout = ops.coo_to_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, torch.from_numpy(w).to(dev))
    plan = ops.build_tile_plan(indptr, indices, n)
    t = timeit(lambda: ops.spmm_csr(indptr, indices, wout, x, n, mean=True, out=out, tiles=plan, force_tiled=True))
    uniq = plan.tile_src.numel() / (n / 32)
    print(f"{name:22s}: {t*1e3:7.1f} us   {2.0*n*f*4/t/1e9:5.2f} TB/s (x + out only)   distinct sources per tile {uniq:.0f}", flush=True)
t = timeit(lambda: out.copy_(x))
print(f"torch copy_           : {t*1e3:7.1f} us   {2.0*n*f*4/t/1e9:5.2f} TB/s")
