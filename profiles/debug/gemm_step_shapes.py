"""The five transform GEMMs of the cfg2 train step (24 495 nodes, F0 = 831, H = 256, q-form), per arithmetic mode."""
import sys
sys.path.insert(0, ".")
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
N = 24495
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x = torch.randn(N, 831, device=dev); w0 = torch.randn(512, 831, device=dev); dz = torch.randn(N, 512, device=dev)
h = torch.randn(N, 256, device=dev); w1 = torch.randn(512, 256, device=dev); dh = torch.randn(N, 256, device=dev)
cases = [("fwd0 NT 24495x512x831", lambda: ops.gemm(x, w0, trans_b=True), 2.0 * N * 512 * 831),
         ("dW0  TN 512x831x24495", lambda: ops.gemm(dz, x, trans_a=True), 2.0 * N * 512 * 831),
         ("fwd1 NT 24495x512x256", lambda: ops.gemm(h, w1, trans_b=True), 2.0 * N * 512 * 256),
         ("dW1  TN 512x256x24495", lambda: ops.gemm(dz, h, trans_a=True), 2.0 * N * 512 * 256),
         ("dX1  NN 24495x256x512", lambda: ops.gemm(dz, w1), 2.0 * N * 512 * 256)]
modes = sys.argv[1:] or ["f32", "split_bf16"]
tot = {m: 0.0 for m in modes}
for name, fn, fl in cases:
    row = []
    for m in modes:
        ops.set_gemm_mode(m)
        us = timeit(fn)
        tot[m] += us
        row.append(f"{m} {us:7.1f} us {fl / us * 1e-6:6.1f} TF")
    print(name, " | ".join(row), flush=True)
print("total", " | ".join(f"{m} {tot[m]:.1f} us" for m in modes))
