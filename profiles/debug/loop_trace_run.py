import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G, distributed as D
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
f0 = 13
dev = torch.device("cuda", 0)
pages = S.make_pages(400, in_feats=f0)
gs = []
for p in pages:
    g = G.PageGraph(p.src, p.dst, p.num_nodes)
    g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
    g.edata["feat"] = torch.from_numpy(p.weight)
    gs.append(g)
res = G.ResidentPages(gs, dev)
sizes = res.page_sizes()
torch.manual_seed(0)
model = gte.GcnSAGE(f0, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
step = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
plan = [ids for ep in range(20) for ids in (r[0] for r in D.plan_epoch(sizes, 100, 1, seed=42, epoch=ep))]
pipe = loop.BatchPipeline(res)
loop.run_steps(step, pipe, plan[:10]); torch.cuda.synchronize()
t0=time.perf_counter(); loop.run_steps(step, pipe, plan[:40]); torch.cuda.synchronize(); print("ms/step", (time.perf_counter()-t0)/40*1e3)
