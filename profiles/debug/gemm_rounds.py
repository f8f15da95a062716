"""How the tile count maps to time: NT GEMMs with 256 / 512 / 768 / 1024 ... tiles (one to four per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gnn_tableextraction_amd import _lib
lib, P, cs = _lib.load(), _lib.ptr, _lib.current_stream
dev = "cuda:0"
def timeit(fn, reps=40):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
W0 = torch.randn(256, 1662, device=dev) * 0.02; b0 = torch.randn(256, device=dev)
W1 = torch.randn(256, 512, device=dev) * 0.02
for M in (4096, 8192, 12288, 16384, 20480, 24576, 28672, 32768, 49152):
    x = torch.randn(M, 831, device=dev); t = torch.empty(M, 512, device=dev)
    us = timeit(lambda: lib.gte_sage_transform_fwd(P(x), 831, 831, P(W0), 1662, P(b0), 256, P(t), 512, M, cs()))
    tiles = (M // 128) * 4
    print(f"L0 fwd K=831 N=512 M={M:6d}: {tiles:5d} tiles of 128x128 = {tiles/256:.2f}/CU  {us:7.1f} us  {2.0*M*831*512/us/1e6:6.1f} TF  ({us/(tiles/256):.1f} us per tile-per-CU)", flush=True)
for M in (8192, 16384, 24576, 32768, 49152):
    h = torch.randn(M, 256, device=dev); ahn = torch.randn(M, 256, device=dev); z = torch.empty(M, 256, device=dev)
    us = timeit(lambda: lib.gte_sage_linear_fwd(P(h), 256, 256, P(ahn), 256, 256, P(W1), 512, P(b0), None, None, 1e-5, 0, None, 0, None, P(z), 256, M, 256, cs()))
    print(f"L1 fwd K=512 N=256 M={M:6d}: {us:7.1f} us  {2.0*M*512*256/us/1e6:6.1f} TF", flush=True)
