"""Split-bf16 GEMM mode against the fp32 MFMA kernel and fp64: every operand layout, ragged shapes, two K segments,
split-K weight gradients.  Error unit: 2^-24 * sum_k |a_k b_k| (one fp32 rounding of the dot product's magnitude)."""
import sys, time
sys.path.insert(0, ".")
import importlib, torch
pkg = importlib.import_module("gnn-tableextraction_amd")
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)


def err_units(c, a64, b64):
    ref = a64 @ b64
    mag = a64.abs() @ b64.abs()
    u = mag * 2.0 ** -24
    e = (c.double().cpu() - ref).abs() / u.clamp_min(1e-300)
    return e.max().item(), e.pow(2).mean().sqrt().item()


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


bad = 0
shapes = [(24495, 512, 831), (24495, 512, 256), (24495, 256, 512), (1000, 256, 77), (300, 130, 1030), (129, 128, 16), (64, 128, 33),
          (5000, 831, 512)]
for (m, n, k) in shapes:
    for ta in (False, True):
        for tb in (False, True):
            a = torch.randn((k, m) if ta else (m, k), generator=g) * (1 + torch.arange(m if ta else k) % 5)
            b = torch.randn((n, k) if tb else (k, n), generator=g) * 0.1
            a64 = (a.t() if ta else a).double(); b64 = (b.t() if tb else b).double()
            ad, bd = a.to(dev), b.to(dev)
            res = {}
            for mode in ("f32", "split_bf16"):
                ops.set_gemm_mode(mode)
                c = ops.gemm(ad, bd, trans_a=ta, trans_b=tb)
                mx, rms = err_units(c, a64, b64)
                us = timeit(lambda: ops.gemm(ad, bd, trans_a=ta, trans_b=tb)) if m >= 5000 else 0.0
                res[mode] = (mx, rms, us)
            flag = "" if res["split_bf16"][0] <= max(1.5 * res["f32"][0], 6.0) else "  <-- BAD"
            bad += bool(flag)
            print(f"M {m:6d} N {n:4d} K {k:5d} {'T' if ta else 'N'}{'T' if tb else 'N'}  f32 max {res['f32'][0]:6.2f} rms {res['f32'][1]:.3f} {res['f32'][2]:7.1f} us |"
                  f" split max {res['split_bf16'][0]:6.2f} rms {res['split_bf16'][1]:.3f} {res['split_bf16'][2]:7.1f} us{flag}", flush=True)

# weight gradient through the split-K path with two K segments: dW = dz^T [x1 | x2]
for (nodes, n_out, k1, k2) in [(24495, 256, 256, 256), (24495, 256, 831, 831), (3000, 256, 100, 60)]:
    dz = torch.randn(nodes, n_out, generator=g) * 0.01
    x1 = torch.randn(nodes, k1, generator=g); x2 = torch.randn(nodes, k2, generator=g)
    ref = dz.double().t() @ torch.cat([x1, x2], 1).double()
    mag = dz.double().abs().t() @ torch.cat([x1, x2], 1).double().abs()
    out = {}
    for mode in ("f32", "split_bf16"):
        ops.set_gemm_mode(mode)
        o = torch.empty(n_out, k1 + k2, device=dev)
        dzd, x1d, x2d = dz.to(dev), x1.to(dev), x2.to(dev)
        ops.sage_linear_dw(dzd, x1d, x2d, o)
        e = ((o.double().cpu() - ref).abs() / (mag * 2.0 ** -24)).max().item()
        us = timeit(lambda: ops.sage_linear_dw(dzd, x1d, x2d, o))
        out[mode] = (e, us)
    flag = "" if out["split_bf16"][0] <= max(1.5 * out["f32"][0], 6.0) else "  <-- BAD"
    bad += bool(flag)
    print(f"dW nodes {nodes} out {n_out} k {k1}+{k2}: f32 max {out['f32'][0]:.2f} {out['f32'][1]:.1f} us | split max {out['split_bf16'][0]:.2f} {out['split_bf16'][1]:.1f} us{flag}", flush=True)
ops.set_gemm_mode("f32")
print("BAD" if bad else "ALL OK", bad)
