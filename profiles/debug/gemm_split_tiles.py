"""Round cost of the split / fp32 GEMM kernels per tile shape (GTE_GEMM_BM=64|128 forces the row tile): time vs K at a
fixed tile count, for the cost model of make_plan."""
import sys, os
sys.path.insert(0, ".")
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
mode = sys.argv[1]
ops.set_gemm_mode(mode)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
bm = int(os.environ.get("GTE_GEMM_BM", "0"))
for m in (128 * 64, 128 * 128, 24495, 128 * 256):
    for n in (256, 512):
        row = []
        for k in (128, 256, 512, 831, 1662):
            a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev)
            row.append(timeit(lambda: ops.gemm(a, b, trans_b=True)))
        print(f"{mode} bm {bm} M {m} N {n}: " + " ".join(f"{x:7.1f}" for x in row), flush=True)
