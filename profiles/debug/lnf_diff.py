"""where gte_gemm_p3_nt_ln_fwd differs from gte_gemm_p3_nt + gte_ln_relu_fwd_p3 (debug aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnn_tableextraction_amd import _lib, ops
DEV = "cuda:0"
lib, P = _lib.load(), _lib.ptr
for m, n, k, relu in [(3000, 218, 13, True), (3000, 218, 13, False), (3000, 256, 13, True), (3000, 96, 363, True)]:
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    x1, x2 = torch.randn(m, k, device=DEV, generator=g), torch.randn(m, k, device=DEV, generator=g)
    kp, ld = -(-k // 16) * 16, -(-n // 16) * 16
    wb = torch.zeros(n, 2 * kp, device=DEV)
    wb[:, :k], wb[:, kp:kp + k] = torch.randn(n, k, device=DEV, generator=g) * 0.1, torch.randn(n, k, device=DEV, generator=g) * 0.1
    w = ops.P3(ops.p3_from_f32(wb).data, n, 2 * kp)
    bias, gamma, beta = (torch.randn(n, device=DEV, generator=g) for _ in range(3))
    a1, a2 = ops.p3_from_f32(x1), ops.p3_from_f32(x2)
    z0 = torch.zeros((m, ld), device=DEV); y0 = torch.zeros((m, ld), device=DEV); st0 = torch.zeros(2 * m, device=DEV)
    yp0 = ops.P3.empty(m, n, DEV); yp0.data.zero_()
    ops.gemm_p3_nt(a1, w, a2=a2, bias=bias, out=z0[:, :n])
    _lib.check(lib.gte_ln_relu_fwd_p3(P(z0), ld, P(gamma), P(beta), 1e-5, int(relu), P(y0), ld, P(yp0.data), yp0.ldp, P(st0), m, n,
                                      _lib.current_stream()), "x")
    z1 = torch.zeros((m, ld), device=DEV); y1 = torch.zeros((m, ld), device=DEV); st1 = torch.zeros(2 * m, device=DEV)
    yp1 = ops.P3.empty(m, n, DEV); yp1.data.zero_()
    ops.gemm_p3_nt_ln_fwd(a1, w, a2, bias, gamma, beta, 1e-5, relu, z1, y=y1, yp3=yp1, stats=st1)
    d = (y1 - y0).abs()
    idx = torch.nonzero(d > 0)
    print(m, n, k, relu, "z", torch.equal(z1, z0), "st", torch.equal(st1, st0), "y", torch.equal(y1, y0), "img", torch.equal(yp1.data, yp0.data),
          "ndiff", idx.shape[0], "max", float(d.max()), "first", idx[:6].tolist(),
          [(float(y0[i, j]), float(y1[i, j])) for i, j in idx[:4].tolist()])
