"""Achievable HBM bandwidth of a plain device copy at the sizes of the step's HBM-bound kernels (yardstick).
usage: python profiles/copy_bw.py"""
import torch
dev = "cuda:0"
for mb in (6, 25, 50, 100, 400, 2000):
    n = mb * 1024 * 1024 // 4
    x, y = torch.randn(n, device=dev), torch.empty(n, device=dev)
    for _ in range(20): y.copy_(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 100
    s.record()
    for _ in range(reps): y.copy_(x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"copy {mb:5d} MB (read + write {2 * mb} MB): {ms * 1e3:8.1f} us  {2 * mb / 1024 / ms * 1e3 / 1e3:6.2f} TB/s", flush=True)
