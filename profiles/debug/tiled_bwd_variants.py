"""cfg4 aggregation forward (in-edge CSR, mean) and backward (out-edge CSR, sum) with several builds of the library in ONE
process: interleaved rounds, median.  usage: python tiled_bwd_variants.py lib_a.so lib_b.so ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_tableextraction_amd import _lib, ops
from gnn_tableextraction_amd.data import synthetic as S
dev = "cuda:0"
n, f = 1_000_000, 512
src, dst, w = S.make_knn_stress_graph(n, 12)
dst_t, src_t, w_t = torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), torch.from_numpy(w).to(dev)
indptr, indices, perm, wout = ops.coo_to_csr(dst_t, src_t, n, w_t)
plan = ops.build_tile_plan(indptr, indices, n)
w_bwd = w_t * ops.inv_degree(indptr)[dst_t.long()]
r_indptr, r_indices, _, r_w = ops.coo_to_csr(src_t, dst_t, n, w_bwd)
r_plan = ops.build_tile_plan(r_indptr, r_indices, n)
x = torch.randn(n, f, device=dev); out = torch.empty_like(x)
libs = {"base": _lib.load()}
for p in sys.argv[1:]:
    l = ctypes.CDLL(os.path.abspath(p))
    l.gte_spmm_csr_tiled.restype, l.gte_spmm_csr_tiled.argtypes = _lib.SIGNATURES["gte_spmm_csr_tiled"]
    libs[os.path.basename(p)] = l
P, st = _lib.ptr, _lib.current_stream()
def call(l, o, bwd):
    if bwd:
        rc = l.gte_spmm_csr_tiled(P(r_indptr), P(r_indices), P(r_plan.local_index), P(r_w), P(r_plan.tile_ptr), P(r_plan.tile_src), P(x), f, P(o), f, n, f, 0, 0, st)
    else:
        rc = l.gte_spmm_csr_tiled(P(indptr), P(indices), P(plan.local_index), P(wout), P(plan.tile_ptr), P(plan.tile_src), P(x), f, P(o), f, n, f, 1, 0, st)
    assert rc == 0
alg = 2.0 * n * f * 4 + 8.0 * len(src) + 4.0 * (n + 1)
for bwd in (False, True):
    ref = torch.empty_like(x)
    call(libs["base"], ref, bwd)
    times = {k: [] for k in libs}
    for k, l in libs.items():
        out.zero_(); call(l, out, bwd); torch.cuda.synchronize()
        print("bwd" if bwd else "fwd", k, "bitwise equal to base:", bool(torch.equal(out, ref)), flush=True)
    del ref
    for rnd in range(6):
        for k, l in libs.items():
            for _ in range(3): call(l, out, bwd)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): call(l, out, bwd)
            e.record(); torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e) / 10)
    for k, t in times.items():
        m = float(np.median(t))
        print(f"{'bwd' if bwd else 'fwd'} {k:20s}: median {m*1e3:7.1f} us  min {min(t)*1e3:7.1f}  {alg/m/1e9:6.2f} TB/s = {alg/m/1e9/8:5.3f} of 8 TB/s", flush=True)
