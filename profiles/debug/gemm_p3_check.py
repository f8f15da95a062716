"""Planes GEMMs (csrc/gemm_p3.hip): correctness against the split GEMM (bitwise) / fp64 and timings on the step's shapes."""
import sys, os
sys.path.insert(0, ".")
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)

def timeit(fn, n=200):
    for _ in range(2000): fn()          # sustained clocks
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def relerr(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())

# ---- round trip
for (r, c) in ((5, 3), (300, 831), (1000, 256), (77, 16), (64, 17)):
    x = torch.randn(r, c, device=dev) * torch.exp(torch.randn(r, c, device=dev) * 3)
    img = ops.p3_from_f32(x)
    back = ops.p3_to_f32(img)
    xt = ops.p3_to_f32(ops.p3_from_f32(x, transpose=True))
    print(f"roundtrip {r}x{c}: exact {bool((back == x).all())} transposed exact {bool((xt == x.t()).all())}", flush=True)

# ---- NT: one K segment
ops.set_gemm_mode("split_bf16")
for (m, n, k) in ((100, 128, 16), (128, 128, 32), (300, 512, 831), (2000, 256, 256), (24437, 512, 831), (24437, 512, 256), (777, 200, 50)):
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); bias = torch.randn(n, device=dev)
    ref_split = ops.gemm(a, b, trans_b=True)
    c = ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b))
    ref64 = a.double() @ b.double().t()
    print(f"NT {m}x{n}x{k}: bitwise == split kernel {bool((c == ref_split).all())}  rel err vs fp64 {relerr(c, ref64):.2e} "
          f"(split {relerr(ref_split, ref64):.2e})  max|diff| {float((c - ref_split).abs().max()):.3e}", flush=True)
    cb = ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b), bias=bias, bias_cols=n // 2, relu=True)
    refb = (ref_split + torch.cat([bias[: n // 2], torch.zeros(n - n // 2, device=dev)])).clamp_min(0)
    print(f"   bias/relu max diff {float((cb - refb).abs().max()):.3e}", flush=True)
# integer operands: exact
a = torch.randint(-8, 9, (1000, 100), device=dev).float(); b = torch.randint(-8, 9, (300, 100), device=dev).float()
c = ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b))
print("NT integers exact:", bool((c.double() == a.double() @ b.double().t()).all()), flush=True)
# ---- NT: two K segments (dX shape)
for (m, n, k1, k2) in ((500, 256, 256, 256), (24437, 256, 256, 256), (333, 100, 40, 24)):
    a1 = torch.randn(m, k1, device=dev); a2 = torch.randn(m, k2, device=dev)
    w = torch.randn(n, k1 + k2, device=dev)
    kb1 = -(-k1 // 16) * 16
    bimg = ops.P3.empty(n, kb1 + k2, dev)
    bimg.data.zero_()
    # rows of b: blocks of segment 1 then blocks of segment 2
    full = torch.zeros(n, kb1 + k2, device=dev); full[:, :k1] = w[:, :k1]; full[:, kb1:] = w[:, k1:]
    bimg = ops.p3_from_f32(full)
    c = ops.gemm_p3_nt(ops.p3_from_f32(a1), bimg, a2=ops.p3_from_f32(a2))
    ref64 = torch.cat([a1, a2], 1).double() @ w.double().t()
    print(f"NT2 {m}x{n}x({k1}+{k2}): rel err vs fp64 {relerr(c, ref64):.2e}", flush=True)
# ---- TN
for (m, n, k, two) in ((256, 256, 64, False), (256, 831, 1000, False), (256, 256, 24437, True), (256, 831, 24437, True), (100, 50, 333, False), (218, 63, 5000, True)):
    a = torch.randn(k, m, device=dev); b = torch.randn(k, n, device=dev); a2 = torch.randn(k, m, device=dev)
    if two:
        c = ops.gemm_p3_tn(ops.p3_from_f32(a), ops.p3_from_f32(b), a2=ops.p3_from_f32(a2), two_segments=True)
        ref64 = torch.cat([a.double().t() @ b.double(), a2.double().t() @ b.double()], 1)
    else:
        c = ops.gemm_p3_tn(ops.p3_from_f32(a), ops.p3_from_f32(b))
        ref64 = a.double().t() @ b.double()
    ref_split = ops.gemm(a, b, trans_a=True)
    print(f"TN {m}x{n}x{k} two={two}: rel err vs fp64 {relerr(c, ref64):.2e} (split kernel {relerr(ref_split, ref64[:, :n]):.2e})", flush=True)
a = torch.randint(-8, 9, (1003, 200), device=dev).float(); b = torch.randint(-8, 9, (1003, 300), device=dev).float()
c = ops.gemm_p3_tn(ops.p3_from_f32(a), ops.p3_from_f32(b))
print("TN integers exact:", bool((c.double() == a.double().t() @ b.double()).all()), flush=True)

# ---- timings on the step's shapes (operands pre-split)
n_nodes = 24437
def gf(m, n, k): return 2.0 * m * n * k * 1e-9
X = ops.p3_from_f32(torch.randn(n_nodes, 831, device=dev)); W0 = ops.p3_from_f32(torch.randn(512, 831, device=dev))
H = ops.p3_from_f32(torch.randn(n_nodes, 256, device=dev)); W1 = ops.p3_from_f32(torch.randn(512, 256, device=dev))
DZ = ops.p3_from_f32(torch.randn(n_nodes, 256, device=dev)); Q = ops.p3_from_f32(torch.randn(n_nodes, 256, device=dev))
W1T = ops.p3_from_f32(torch.randn(256, 512, device=dev))
out512 = torch.empty(n_nodes, 512, device=dev); out256 = torch.empty(n_nodes, 256, device=dev)
dw0 = torch.empty(256, 1662, device=dev); dw1 = torch.empty(256, 512, device=dev)
xf = torch.randn(n_nodes, 831, device=dev); wf = torch.randn(512, 831, device=dev)
for name, fn, g in (
    ("L0 fwd  NT 24437x512x831", lambda: ops.gemm_p3_nt(X, W0, out=out512), gf(n_nodes, 512, 831)),
    ("L1 fwd  NT 24437x512x256", lambda: ops.gemm_p3_nt(H, W1, out=out512), gf(n_nodes, 512, 256)),
    ("dX      NT 24437x256x(256+256)", lambda: ops.gemm_p3_nt(DZ, W1T, a2=Q, out=out256), gf(n_nodes, 256, 512)),
    ("dW1     TN 256x(256+256)x24437", lambda: ops.gemm_p3_tn(DZ, H, a2=Q, two_segments=True, out=dw1), gf(256, 512, n_nodes)),
    ("dW0     TN 256x(831+831)x24437", lambda: ops.gemm_p3_tn(DZ, X, a2=Q, two_segments=True, out=dw0), gf(256, 1662, n_nodes)),
    ("split kernel L0 fwd", lambda: ops.gemm(xf, wf, trans_b=True, out=out512), gf(n_nodes, 512, 831)),
):
    us = timeit(fn)
    print(f"{name:36s} {us:8.1f} us  {g / us * 1e3:7.1f} TF fp32-eq  ({6 * g / us * 1e3:7.1f} TF bf16)", flush=True)
