"""gte_gemm_p3_nt_ln_bwd (dX with the LayerNorm backward epilogue) against the two launches it replaces, at the step's shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
_lib = importlib.import_module("gnn-tableextraction_amd._lib")
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, f = 24437, 256
DZ, Q = ops.p3_from_f32(torch.randn(n, f, device=dev)), ops.p3_from_f32(torch.randn(n, f, device=dev))
W = ops.p3_from_f32(torch.randn(f, 2 * f, device=dev) / 16)
t = torch.randn(n, 2 * f, device=dev); z = t[:, :f]
stats = torch.cat([z.mean(1), 1 / torch.sqrt(z.var(1, unbiased=False) + 1e-5)]).contiguous()
g, be = torch.ones(f, device=dev), torch.zeros(f, device=dev)
dy, dz, dzp = torch.empty(n, f, device=dev), torch.empty(n, f, device=dev), ops.P3.empty(n, f, dev)
dg, db, dbias = (torch.zeros(f, device=dev) for _ in range(3))
lib, P = _lib.load(), _lib.ptr
ws = torch.empty(int(lib.gte_ln_relu_bwd_workspace_bytes(n, f)), dtype=torch.uint8, device=dev)
def two():
    ops.gemm_p3_nt(DZ, W, a2=Q, out=dy)
    lib.gte_ln_relu_bwd_p3(P(dy), f, P(z), 2 * f, P(stats), P(g), P(be), 1, P(dy), f, P(dzp.data), dzp.ldp, P(dg), P(db), P(dbias), n, f, P(ws), ws.numel(), _lib.current_stream())
def one():
    ops.gemm_p3_nt_ln_bwd(DZ, W, Q, z, stats, g, be, True, dz, dzp, dg, db, dbias)
lib.gte_fold_defer_begin(None)
for name, fn in (("two launches", two), ("one launch", one), ("two launches", two), ("one launch", one)):
    for _ in range(1500): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / 300:7.1f} us", flush=True)
