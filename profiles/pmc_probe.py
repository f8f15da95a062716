"""Run ONE kernel shape a few times (for rocprofv3 --pmc passes).  usage: python3 profiles/pmc_probe.py <nt|tn|nn|spmm|spmm256> [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import ops
dev = "cuda:0"
what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M = 24495
if what == "nt":
    a1, a2 = torch.randn(M, 831, device=dev), torch.randn(M, 831, device=dev)
    w, b = torch.randn(256, 1662, device=dev) * 0.02, torch.randn(256, device=dev)
    fn = lambda: ops.sage_linear_fwd(a1, a2, w, b, None, None, 1e-5, False, False)
elif what == "tn":
    dz, x = torch.randn(M, 256, device=dev), torch.randn(M, 831, device=dev)
    out = torch.empty(256, 831, device=dev)
    fn = lambda: ops.gemm(dz, x, trans_a=True, out=out)
elif what == "nn":
    dz, w = torch.randn(M, 256, device=dev), torch.randn(256, 512, device=dev)
    out = torch.empty(M, 256, device=dev)
    fn = lambda: ops.gemm(dz, w[:, :256], out=out)
elif what == "spmm":
    from gnn_tableextraction_amd.data import synthetic as S
    n, k, f = 1_000_000, 12, 512
    src, dst, w = S.make_knn_stress_graph(n, k)
    indptr, indices, perm, wout = ops.coo_to_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, torch.from_numpy(w).to(dev))
    x = torch.randn(n, f, device=dev); out = torch.empty_like(x)
    plan = ops.build_tile_plan(indptr, indices, n) if os.environ.get("GTE_TILED", "1") == "1" else None
    fn = lambda: ops.spmm_csr(indptr, indices, wout, x, n, mean=True, out=out, tiles=plan, force_tiled=True)
if what == "spmm256":                      # the 256-wide aggregation of a cfg2 batch (L2-resident regime)
    from gnn_tableextraction_amd.data import synthetic as S
    pages = S.make_pages(100, in_feats=13)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    N = int(off[-1])
    graph = gte.PageGraph(src, dst, N, device=dev)
    csr = graph.in_csr(); wt = graph.in_weights(torch.from_numpy(w).to(dev))
    x = torch.randn(N, 256, device=dev); out = torch.empty_like(x)
    fn = lambda: ops.spmm_csr(csr.indptr, csr.indices, wt, x, N, mean=True, out=out)
for _ in range(reps):
    fn()
torch.cuda.synchronize()
