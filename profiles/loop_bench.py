"""Throughput of the REAL training-loop shape: a different 100-page batch every step, built on the device from
resident pages (gte_batch_csr/rows) and run through the eager fused step (no HIP-graph replay possible).
usage: python profiles/loop_bench.py [in_feats] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G, distributed as D
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
f0 = int(sys.argv[1]) if len(sys.argv) > 1 else 831
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = "cuda:0"
pages = S.make_pages(600, in_feats=f0)
gs = []
for p in pages:
    g = G.PageGraph(p.src, p.dst, p.num_nodes)
    g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
    g.edata["feat"] = torch.from_numpy(p.weight)
    gs.append(g)
t0 = time.time(); res = G.ResidentPages(gs, dev); torch.cuda.synchronize(); print(f"resident dataset: {res.n_nodes} nodes, built in {time.time()-t0:.2f}s")
torch.manual_seed(0)
model = gte.GcnSAGE(f0, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
step = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
sizes = res.page_sizes()
plan = [ids for ep in range(20) for ids in (r[0] for r in D.plan_epoch(sizes, 100, 1, seed=42, epoch=ep))]
for ids in plan[:5]:
    bg = res.batch(ids); step.step(bg, bg.ndata["label"])
torch.cuda.synchronize()
nodes = 0; tb = 0.0
t0 = time.perf_counter()
for ids in plan[5:5 + steps]:
    t1 = time.perf_counter()
    bg = res.batch(ids)
    tb += time.perf_counter() - t1
    out3 = step.step(bg, bg.ndata["label"])
    nodes += bg.num_nodes()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"F0={f0}: {steps} steps, {nodes/dt/1e6:.2f} M nodes/s, {dt/steps*1e3:.3f} ms/step (host time in batch(): {tb/steps*1e3:.3f} ms/step), loss {float(out3[0]):.4f}")
