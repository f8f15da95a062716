"""Planes NT GEMM at row counts beyond one round of tiles: time per launch for the tile configuration in GTE_P3_NT_CFG (read
once per process by the library; unset = the chooser).   python profiles/gemm_p3_big_m.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_tableextraction_amd import ops
dev = torch.device("cuda", 0)
cfg = os.environ.get("GTE_P3_NT_CFG", "auto")
g = torch.Generator(device="cpu").manual_seed(1)
for (m, n, k) in [(24437, 512, 831), (36475, 512, 831), (48781, 512, 831), (80149, 512, 831), (486102, 512, 831), (486102, 512, 256),
                  (24294, 2016, 831), (24294, 2016, 1000), (80149, 512, 256)]:
    a = ops.p3_from_f32(torch.randn(m, k, generator=g).relu_().to(dev))       # post-ReLU-like operand: half zeros
    b = ops.p3_from_f32((torch.randn(n, k, generator=g) * 0.05).to(dev))
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    for _ in range(3):
        ops.gemm_p3_nt(a, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.gemm_p3_nt(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"cfg {cfg:>4} NT {m:7d} x {n:4d} x {k:4d}: {ms * 1e3:9.1f} us  {2.0 * m * n * k / ms / 1e9:7.1f} TF fp32-eq", flush=True)
    del a, b, out
