"""Layer-0 forward shape (K = 831, N = 512) with and without the slot-round tail split, interleaved in one process
(GTE_TAIL_SLOTS is read once per process: two libraries are loaded side by side instead)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_tableextraction_amd import _lib
lib, P, cs = _lib.load(), _lib.ptr, _lib.current_stream
dev = "cuda:0"
W0 = torch.randn(256, 1662, device=dev) * 0.02; b0 = torch.randn(256, device=dev)
tail = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=dev)
def timeit(fn, reps=30):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for M in (24495, 24576, 22000, 26500, 20480, 28672):
    x = torch.randn(M, 831, device=dev); t = torch.empty(M, 512, device=dev); t2 = torch.empty(M, 512, device=dev)
    res = {}
    for rnd in range(3):
        for name, on in (("split", True), ("plain", False)):
            lib.gte_gemm_set_tail_workspace(P(tail) if on else None, tail.numel() if on else 0)
            us = timeit(lambda: lib.gte_sage_transform_fwd(P(x), 831, 831, P(W0), 1662, P(b0), 256, P(t if on else t2), 512, M, cs()))
            res.setdefault(name, []).append(us)
    err = float((t - t2).abs().max())
    tiles = -(-M // 128) * 4
    print(f"M={M:6d} ({tiles} tiles, {tiles/256:.2f}/CU): tail workspace on {np.median(res['split']):7.1f} us   off {np.median(res['plain']):7.1f} us   max|diff| {err:.2e}", flush=True)
