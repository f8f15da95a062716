"""NT planes GEMMs of the step alone (time per launch); GTE_LIB_PATH selects an ablation build."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)

def timeit(fn, n=300):
    for _ in range(2000): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

n = 24437
X, W0 = ops.p3_from_f32(torch.randn(n, 831, device=dev)), ops.p3_from_f32(torch.randn(512, 831, device=dev))
H, W1 = ops.p3_from_f32(torch.randn(n, 256, device=dev)), ops.p3_from_f32(torch.randn(512, 256, device=dev))
DZ, Q, W1T = ops.p3_from_f32(torch.randn(n, 256, device=dev)), ops.p3_from_f32(torch.randn(n, 256, device=dev)), ops.p3_from_f32(torch.randn(256, 512, device=dev))
o512, o256 = torch.empty(n, 512, device=dev), torch.empty(n, 256, device=dev)
tag = os.path.basename(os.environ.get("GTE_LIB_PATH", "default"))
for name, fn, gf in (("L0 fwd", lambda: ops.gemm_p3_nt(X, W0, out=o512), 2e-9 * n * 512 * 831),
                     ("L1 fwd", lambda: ops.gemm_p3_nt(H, W1, out=o512), 2e-9 * n * 512 * 256),
                     ("dX", lambda: ops.gemm_p3_nt(DZ, W1T, a2=Q, out=o256), 2e-9 * n * 256 * 512)):
    us = timeit(fn)
    print(f"[{tag}] {name}: {us:7.1f} us  {gf / us * 1e3:6.1f} TF fp32-eq", flush=True)
