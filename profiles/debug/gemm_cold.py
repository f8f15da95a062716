"""The step's GEMM shapes timed one launch at a time, inputs warm (same launch repeated: operands live in the Infinity Cache)
vs cold (512 MB written to another buffer before every launch, as ~700 MB of other traffic separate two uses inside a step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import _lib
lib, P, cs = _lib.load(), _lib.ptr, _lib.current_stream
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24495
x = torch.randn(M, 831, device=dev); W0 = torch.randn(256, 1662, device=dev) * 0.02; b0 = torch.randn(256, device=dev)
t = torch.empty(M, 512, device=dev)
h = torch.randn(M, 256, device=dev); ahn = torch.randn(M, 256, device=dev); W1 = torch.randn(256, 512, device=dev) * 0.02
z = torch.empty(M, 256, device=dev)
dz = torch.randn(M, 256, device=dev); q = torch.randn(M, 256, device=dev); dx = torch.empty(M, 256, device=dev)
gW0 = torch.empty(256, 1662, device=dev); gW1 = torch.empty(256, 512, device=dev)
ws = torch.empty(int(max(lib.gte_sage_qform_dw_workspace_bytes(256, 831, M), lib.gte_sage_qform_dw_workspace_bytes(256, 256, M))), dtype=torch.uint8, device=dev)
tail = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=dev)
lib.gte_gemm_set_tail_workspace(P(tail), tail.numel())
flush = torch.empty(128 << 20, device=dev)          # 512 MB
cases = {
    "L0 fwd  transform  [M,831]x[831,512]": (lambda: lib.gte_sage_transform_fwd(P(x), 831, 831, P(W0), 1662, P(b0), 256, P(t), 512, M, cs()), 2.0 * M * 831 * 512),
    "L1 fwd  linear     [M,512]x[512,256]": (lambda: lib.gte_sage_linear_fwd(P(h), 256, 256, P(ahn), 256, 256, P(W1), 512, P(b0), None, None, 1e-5, 0, None, 0, None, P(z), 256, M, 256, cs()), 2.0 * M * 512 * 256),
    "L1 dX   qform_dx   [M,512]x[512,256]": (lambda: lib.gte_sage_qform_dx(P(dz), 256, P(q), 256, P(W1), 512, 256, 256, P(dx), 256, M, cs()), 2.0 * M * 512 * 256),
    "L1 dW   qform_dw   [256,M]x[M,512]  ": (lambda: lib.gte_sage_qform_dw(P(dz), 256, P(q), 256, P(h), 256, 256, P(gW1), 512, 256, M, P(ws), ws.numel(), cs()), 2.0 * M * 512 * 256),
    "L0 dW   qform_dw   [256,M]x[M,1662] ": (lambda: lib.gte_sage_qform_dw(P(dz), 256, P(q), 256, P(x), 831, 831, P(gW0), 1662, 256, M, P(ws), ws.numel(), cs()), 2.0 * M * 1662 * 256),
}
for name, (fn, flops) in cases.items():
    for _ in range(30): fn()
    torch.cuda.synchronize()
    res = {}
    for mode in ("warm", "cold"):
        evs = []
        for _ in range(20):
            if mode == "cold": flush.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        res[mode] = d[len(d) // 2]
    print(f"{name}: warm {res['warm']:6.1f} us ({flops / res['warm'] / 1e6:5.1f} TF)   cold {res['cold']:6.1f} us ({flops / res['cold'] / 1e6:5.1f} TF)", flush=True)
