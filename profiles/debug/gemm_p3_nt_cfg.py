"""NT planes GEMM under one forced ring configuration (GTE_P3_NT_CFG): bitwise check against the split GEMM + timings."""
import sys, os
sys.path.insert(0, ".")
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = os.environ.get("GTE_P3_NT_CFG", "auto")
def timeit(fn, n=200):
    for _ in range(2000): fn()          # ~0.2 s of back-to-back launches first: sustained clocks
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
ops.set_gemm_mode("split_bf16")
ok = True
for (m, n, k) in ((100, 128, 16), (300, 512, 831), (2000, 256, 256), (24437, 512, 831), (24437, 256, 512), (777, 200, 50), (5000, 512, 48)):
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev)
    ref = a.double() @ b.double().t()
    c = ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b))
    c2 = ops.gemm(a, b, trans_b=True)
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    same = bool((c == c2).all())
    ok = ok and err < 3e-6
    print(f"cfg {cfg} NT {m}x{n}x{k}: rel err {err:.2e} bitwise==split {same}", flush=True)
m, n, k1, k2 = 24437, 256, 256, 256
a1 = torch.randn(m, k1, device=dev); a2 = torch.randn(m, k2, device=dev); w = torch.randn(n, k1 + k2, device=dev)
c = ops.gemm_p3_nt(ops.p3_from_f32(a1), ops.p3_from_f32(w), a2=ops.p3_from_f32(a2))
ref = torch.cat([a1, a2], 1).double() @ w.double().t()
print(f"cfg {cfg} NT2: rel err {float((c.double() - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
n_nodes = 24437
def gf(m, n, k): return 2.0 * m * n * k * 1e-9
X = ops.p3_from_f32(torch.randn(n_nodes, 831, device=dev)); W0 = ops.p3_from_f32(torch.randn(512, 831, device=dev))
H = ops.p3_from_f32(torch.randn(n_nodes, 256, device=dev)); W1 = ops.p3_from_f32(torch.randn(512, 256, device=dev))
DZ = ops.p3_from_f32(torch.randn(n_nodes, 256, device=dev)); Q = ops.p3_from_f32(torch.randn(n_nodes, 256, device=dev))
W1T = ops.p3_from_f32(torch.randn(256, 512, device=dev))
out512 = torch.empty(n_nodes, 512, device=dev); out256 = torch.empty(n_nodes, 256, device=dev)
for name, fn, g in (
    ("L0 fwd  NT 24437x512x831", lambda: ops.gemm_p3_nt(X, W0, out=out512), gf(n_nodes, 512, 831)),
    ("L1 fwd  NT 24437x512x256", lambda: ops.gemm_p3_nt(H, W1, out=out512), gf(n_nodes, 512, 256)),
    ("dX      NT 24437x256x(256+256)", lambda: ops.gemm_p3_nt(DZ, W1T, a2=Q, out=out256), gf(n_nodes, 256, 512)),
):
    us = timeit(fn)
    print(f"cfg {cfg} {name:34s} {us:8.1f} us  {g / us * 1e3:7.1f} TF fp32-eq  ({6 * g / us * 1e3:7.1f} TF bf16)", flush=True)
for mm in (20000, 22000, 26000, 30000):
    Xm = ops.p3_from_f32(torch.randn(mm, 831, device=dev)); o = torch.empty(mm, 512, device=dev)
    us = timeit(lambda: ops.gemm_p3_nt(Xm, W0, out=o))
    print(f"cfg {cfg} L0 fwd M={mm}: {us:8.1f} us {gf(mm, 512, 831) / us * 1e3:7.1f} TF fp32-eq", flush=True)
