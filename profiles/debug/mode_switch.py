"""Debug: the train loop before / after switching the GEMM mode inside one process (fused Adam on / off)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G, distributed as D, ops
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
dev = torch.device("cuda", 0)
pages = S.make_pages(int(os.environ.get("PAGES", "300")), in_feats=831)
gs = []
for p in pages:
    g = G.PageGraph(p.src, p.dst, p.num_nodes)
    g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
    g.edata["feat"] = torch.from_numpy(p.weight)
    gs.append(g)
res = G.ResidentPages(gs, dev)
torch.manual_seed(42)
model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
pipe = loop.BatchPipeline(res)
plans = [[r[0] for r in D.plan_epoch(res.page_sizes(), 100, 1, seed=42, epoch=e)] for e in range(60)]
def phase(name, epochs):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for pl in epochs:
        loop.run_steps(tr, pipe, pl); n += len(pl)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step  fused_steps={tr.adam_fused_steps} t={tr.t} dev_host={tr._step_dev_host} dev={int(tr._step_dev.item())}", flush=True)
phase("warm f32", plans[:3]); phase("f32", plans[3:13])
ops.set_gemm_mode("split_bf16")
phase("warm split", plans[13:16]); phase("split", plans[16:26])
ops.set_gemm_mode("f32")
phase("warm f32 again", plans[26:29]); phase("f32 again", plans[29:39])
# --- what bench.py does between its headline region and the other-mode run ---
print("kernel timers phase"); ops.enable_kernel_timers(True); phase("timers", plans[0:1]); kt = ops.kernel_timer_report(); ops.enable_kernel_timers(False)
phase("f32 after timers", plans[1:6])
x = torch.randn(2048, 831, device=dev); w = torch.randn(512, 831, device=dev)
for m in ("f32", "split_bf16"):
    ops.set_gemm_mode(m); ops.gemm(x, w, trans_b=True)
ops.set_gemm_mode("split_bf16")
phase("split after gemm probe", plans[6:12])
ops.set_gemm_mode("f32")
phase("f32 after gemm probe", plans[12:18])

ops.set_gemm_mode("split_bf16")
phase("split 20 epochs back to back", plans[18:38])
ops.set_gemm_mode("f32")
phase("f32 20 epochs back to back", plans[38:58])
