"""cProfile of the host side of the train loop (240 steps in epochs of 12, as bench.py runs them): where the ~190 us of host
time per step go."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
from gnn_tableextraction_amd.models import loop as L
dev = torch.device("cuda:0")
NP = 600
pages = S.make_pages(NP, in_feats=831)
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
pipe = L.BatchPipeline(res)
rng = np.random.default_rng(0)
plans = [[rng.choice(NP, 100, replace=False) for _ in range(12)] for _ in range(22)]
for pl in plans[:2]: L.run_steps(tr, pipe, pl)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for pl in plans[2:]: L.run_steps(tr, pipe, pl)
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"240 steps under cProfile: host {t_host / 240 * 1e6:.0f} us per step, with the device {t_all / 240 * 1e6:.0f} us per step")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
