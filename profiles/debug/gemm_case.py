"""ONE of the step's GEMM launches, a few times (for rocprofv3 --pmc passes): python gemm_case.py <l0fwd|l1fwd|l1dx|l1dw|l0dw> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnn_tableextraction_amd import _lib
lib, P, cs = _lib.load(), _lib.ptr, _lib.current_stream
dev = "cuda:0"
what, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
M = 24495
x = torch.randn(M, 831, device=dev); W0 = torch.randn(256, 1662, device=dev) * 0.02; b0 = torch.randn(256, device=dev)
t = torch.empty(M, 512, device=dev)
h = torch.randn(M, 256, device=dev); ahn = torch.randn(M, 256, device=dev); W1 = torch.randn(256, 512, device=dev) * 0.02
z = torch.empty(M, 256, device=dev)
dz = torch.randn(M, 256, device=dev); q = torch.randn(M, 256, device=dev); dx = torch.empty(M, 256, device=dev)
gW0 = torch.empty(256, 1662, device=dev); gW1 = torch.empty(256, 512, device=dev)
ws = torch.empty(int(max(lib.gte_sage_qform_dw_workspace_bytes(256, 831, M), lib.gte_sage_qform_dw_workspace_bytes(256, 256, M))), dtype=torch.uint8, device=dev)
tail = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=dev)
lib.gte_gemm_set_tail_workspace(P(tail), tail.numel())
fn = {
    "l0fwd": lambda: lib.gte_sage_transform_fwd(P(x), 831, 831, P(W0), 1662, P(b0), 256, P(t), 512, M, cs()),
    "l1fwd": lambda: lib.gte_sage_linear_fwd(P(h), 256, 256, P(ahn), 256, 256, P(W1), 512, P(b0), None, None, 1e-5, 0, None, 0, None, P(z), 256, M, 256, cs()),
    "l1dx": lambda: lib.gte_sage_qform_dx(P(dz), 256, P(q), 256, P(W1), 512, 256, 256, P(dx), 256, M, cs()),
    "l1dw": lambda: lib.gte_sage_qform_dw(P(dz), 256, P(q), 256, P(h), 256, 256, P(gW1), 512, 256, M, P(ws), ws.numel(), cs()),
    "l0dw": lambda: lib.gte_sage_qform_dw(P(dz), 256, P(q), 256, P(x), 831, 831, P(gW0), 1662, 256, M, P(ws), ws.numel(), cs()),
}[what]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
