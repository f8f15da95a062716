import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
f0 = int(sys.argv[1]) if len(sys.argv) > 1 else 831
dev = torch.device("cuda", 0)
pages = S.make_pages(200, in_feats=f0)
gs = []
for p in pages:
    g = G.PageGraph(p.src, p.dst, p.num_nodes)
    g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
    g.edata["feat"] = torch.from_numpy(p.weight)
    gs.append(g)
res = G.ResidentPages(gs, dev)
fixed = [res.batch(list(range(i * 100, i * 100 + 100))) for i in range(2)]
torch.manual_seed(0)
model = gte.GcnSAGE(f0, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
step = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
for i in range(60): step.step(fixed[i % 2], fixed[i % 2].ndata["label"])
torch.cuda.synchronize()
