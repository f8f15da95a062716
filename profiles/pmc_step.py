"""A few steps of the train loop, nothing else (for rocprofv3 --pmc passes: counter collection serialises every dispatch, so
the full bench.py is far too long under it).  usage: python3 profiles/pmc_step.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G, distributed as D
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
pages = S.make_pages(300, in_feats=831)
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
plan = [r[0] for r in D.plan_epoch(res.page_sizes(), 100, 1, seed=42, epoch=0)] + [r[0] for r in D.plan_epoch(res.page_sizes(), 100, 1, seed=42, epoch=1)]
loop.run_steps(tr, pipe, plan[:steps])
torch.cuda.synchronize()
ran = len(pipe)                                  # (the plan may hold fewer steps than asked for)
print("PMC_STEP steps", ran, "nodes_per_step", sum(pipe.nodes(i) for i in range(ran)) / max(ran, 1))
