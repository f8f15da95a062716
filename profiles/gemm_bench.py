"""GEMM / LayerNorm / aggregation micro-benchmark at the cfg2 shapes (run on the GPU box).
usage: python profiles/gemm_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import ops

dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 24495


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def report(name, ms, flops=None, nbytes=None):
    extra = f"{flops / ms / 1e9:8.1f} TF/s" if flops else f"{nbytes / ms / 1e6:8.1f} GB/s"
    print(f"{name:58s} {ms * 1e3:9.1f} us  {extra}", flush=True)


for (k1, k2, n) in [(831, 831, 256), (256, 256, 256), (256, 256, 9), (13, 13, 256), (1000, 1000, 1000)]:
    a1, a2 = torch.randn(M, k1, device=dev), torch.randn(M, k2, device=dev)
    w, b = torch.randn(n, k1 + k2, device=dev) * 0.02, torch.randn(n, device=dev)
    ms = timeit(lambda: ops.sage_linear_fwd(a1, a2, w, b, None, None, 1e-5, False, False))
    report(f"NT fwd  M={M} K={k1}+{k2} N={n}", ms, 2.0 * M * (k1 + k2) * n)
for (mo, n) in [(256, 831), (256, 256), (9, 256), (1000, 1000)]:
    dz, x = torch.randn(M, mo, device=dev), torch.randn(M, n, device=dev)
    out = torch.empty(mo, n, device=dev)
    ms = timeit(lambda: ops.gemm(dz, x, trans_a=True, out=out))
    report(f"TN dW   M={mo} N={n} K={M}", ms, 2.0 * M * mo * n)
for (k, n) in [(256, 256), (9, 256), (1000, 1000)]:
    dz, w = torch.randn(M, k, device=dev), torch.randn(k, 2 * n, device=dev)
    out = torch.empty(M, n, device=dev)
    ms = timeit(lambda: ops.gemm(dz, w[:, :n], out=out))
    report(f"NN dX   M={M} N={n} K={k}", ms, 2.0 * M * k * n)

# LayerNorm fwd / bwd, H = 256
n = 256
z, dy = torch.randn(M, n, device=dev), torch.randn(M, n, device=dev)
g, be = torch.ones(n, device=dev), torch.zeros(n, device=dev)
lib = gte._lib.load()
y, stats = torch.empty_like(z), torch.empty(2 * M, device=dev)
P, cs = gte._lib.ptr, gte._lib.current_stream
ms = timeit(lambda: lib.gte_ln_relu_fwd(P(z), n, P(g), P(be), 1e-5, 1, P(y), n, P(stats), M, n, cs()))
report(f"LN+ReLU fwd  M={M} n={n}", ms, nbytes=2.0 * M * n * 4)
dg, db, dbias = (torch.zeros(n, device=dev) for _ in range(3))
ms = timeit(lambda: ops.ln_relu_bwd(dy, z, stats, g, be, True, dg, db, dbias))
report(f"LN+ReLU bwd (+colsums)  M={M} n={n}", ms, nbytes=3.0 * M * n * 4)

# aggregation at cfg2 sizes
from gnn_tableextraction_amd.data import synthetic as S
pages = S.make_pages(100, in_feats=13)
src, dst, w, feat, label, off = S.concat_pages(pages)
N = int(off[-1])
graph = gte.PageGraph(src, dst, N, device=dev)
csr = graph.in_csr()
wt = graph.in_weights(torch.from_numpy(w).to(dev))
for f in (831, 256, 9):
    x = torch.randn(N, f, device=dev)
    o = torch.empty_like(x)
    ms = timeit(lambda: ops.spmm_csr(csr.indptr, csr.indices, wt, x, N, mean=True, out=o))
    report(f"spmm N={N} E={len(src)} F={f}", ms, nbytes=2.0 * N * f * 4 + 8.0 * len(src) + 4.0 * N)
