"""Per-step device time of bench.py's timed region (W warm-up steps, barrier, K steps): where a short run loses time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.models import loop
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
W, K = 5, 20
pages = bench.make_pages_parallel(1200, 831, 0, 8)
dev = torch.device("cuda", 0)
res = G.ResidentPages(bench.to_page_graphs(gte, pages), dev)
pipe = loop.BatchPipeline(res)
sizes = res.page_sizes()
torch.manual_seed(42)
model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
import gc
PRE = float(os.environ.get("PRE_WARM_S", "0"))
if os.environ.get("NO_GC"): gc.disable()
for trial in range(4):
    if PRE > 0:                                   # settle clocks: the same loop for PRE seconds, then idle briefly
        pre, _ = bench.epoch_steps(sizes, 100, 7, 100 + 10 * trial, int(PRE / 0.00078))
        for plan in pre: loop.run_steps(tr, pipe, plan)
        torch.cuda.synchronize()
    warm, ep = bench.epoch_steps(sizes, 100, 42, 10 * trial, W)
    timed, ep = bench.epoch_steps(sizes, 100, 42, ep, K)
    for plan in warm: loop.run_steps(tr, pipe, plan)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    k = [0]
    t0 = time.perf_counter()
    evs[0].record()
    def on_step(s, g, o):
        k[0] += 1; evs[k[0]].record()
    for plan in timed: loop.run_steps(tr, pipe, plan, on_step=on_step)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(K)]
    print(f"trial {trial}: wall {el / K * 1e3:.3f} ms/step; per-step us:", " ".join(f"{x:.0f}" for x in d), flush=True)
