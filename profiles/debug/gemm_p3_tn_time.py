"""dW GEMMs of the step alone (planes TN kernel, csrc/gemm_p3.hip): time per launch, optionally zero operands (power) and
through a row map.  GTE_P3_TN_SPLITS forces the split count, GTE_LIB_PATH an ablation build."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)
zero = "zero" in sys.argv

def timeit(fn, n=300):
    for _ in range(1500): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

n = 24437
mk = (lambda r, c: torch.zeros(r, c, device=dev)) if zero else (lambda r, c: torch.randn(r, c, device=dev))
X, H = ops.p3_from_f32(mk(n, 831)), ops.p3_from_f32(mk(n, 256))
DZ, Q = ops.p3_from_f32(mk(n, 256)), ops.p3_from_f32(mk(n, 256))
dw0, dw1 = torch.empty(256, 1662, device=dev), torch.empty(256, 512, device=dev)
which = [a for a in sys.argv[1:] if a in ("dw0", "dw1")] or ["dw1", "dw0"]
for name, key, fn, gf in (("dW1 256x(256+256)x24437", "dw1", lambda: ops.gemm_p3_tn(DZ, H, a2=Q, two_segments=True, out=dw1), 2e-9 * 256 * 512 * n),
                          ("dW0 256x(831+831)x24437", "dw0", lambda: ops.gemm_p3_tn(DZ, X, a2=Q, two_segments=True, out=dw0), 2e-9 * 256 * 1662 * n)):
    if key not in which:
        continue
    us = timeit(fn)          # (GEMM + its split-K fold launch; the kernel alone: rocprofv3 --kernel-trace --stats)
    print(f"{name} {'zeros' if zero else 'randn'} splits={os.environ.get('GTE_P3_TN_SPLITS', 'auto')} lib={os.path.basename(os.environ.get('GTE_LIB_PATH', 'default'))}: "
          f"{us:7.1f} us  {gf / us * 1e3:6.1f} TF fp32-eq", flush=True)
