"""Host time per step() call against the device time per step (the one-call step)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
dev = "cuda:0"
pages = S.make_pages(300, in_feats=831)
graphs = []
for p in pages:
    g = gte.PageGraph(p.src, p.dst, p.num_nodes)
    g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
    g.edata["feat"] = torch.from_numpy(p.weight)
    graphs.append(g)
res = G.ResidentPages(graphs, dev)
torch.manual_seed(0)
model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
if tr.wants_p3_features(831): res.enable_p3()
rng = np.random.default_rng(0)
batches = [res.batch(rng.choice(300, 100, replace=False)) for _ in range(6)]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(30): tr.step(batches[i % 6], batches[i % 6].ndata["label"])
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for i in range(120):
        h0 = time.perf_counter()
        tr.step(batches[i % 6], batches[i % 6].ndata["label"])
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
print(f"host {host / 120 * 1e6:.0f} us per step() call, {tot / 120 * 1e6:.0f} us per step with the device")
