"""cfg4 backward aggregation: the bench probe's timing (an event pair per launch through ops.spmm_csr) against one event pair around
ten launches, through ops.spmm_csr and through the C entry point directly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_tableextraction_amd import _lib, ops
from gnn_tableextraction_amd.data import synthetic as S
dev = "cuda:0"
n, f = 1_000_000, 512
src, dst, w = S.make_knn_stress_graph(n, 12)
dst_t, src_t, w_t = torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), torch.from_numpy(w).to(dev)
indptr, indices, perm, wout = ops.coo_to_csr(dst_t, src_t, n, w_t)
w_bwd = w_t * ops.inv_degree(indptr)[dst_t.long()]
r_indptr, r_indices, _, r_w = ops.coo_to_csr(src_t, dst_t, n, w_bwd)
r_plan = ops.build_tile_plan(r_indptr, r_indices, n)
x = torch.randn(n, f, device=dev); out = torch.empty_like(x)
lib, P, st = _lib.load(), _lib.ptr, _lib.current_stream()
def via_ops(): ops.spmm_csr(r_indptr, r_indices, r_w, x, n, mean=False, out=out, tiles=r_plan, force_tiled=True)
def via_c(): lib.gte_spmm_csr_tiled(P(r_indptr), P(r_indices), P(r_plan.local_index), P(r_w), P(r_plan.tile_ptr), P(r_plan.tile_src), P(x), f, P(out), f, n, f, 0, 0, st)
for name, fn in (("ops.spmm_csr", via_ops), ("C entry", via_c)) * 3:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    per = float(np.mean([s.elapsed_time(e) for s, e in evs]))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name}: event pair per launch {per * 1e3:.1f} us, one pair around ten launches {s.elapsed_time(e) * 100:.1f} us per launch")
