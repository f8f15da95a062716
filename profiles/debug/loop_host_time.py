"""How far the host runs ahead of the device in the train loop: host seconds per loop iteration (no synchronisation inside) against
the device time per step; and the C call's share."""
import sys, os, time
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
DIST = os.environ.get("LOOP_DIST") == "1"        # the data-parallel step through a one-rank RCCL process group
if DIST:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4, distributed=DIST)
pipe = L.BatchPipeline(res)
rng = np.random.default_rng(0)
def plan(nsteps):
    return [rng.choice(NP, 100, replace=False) for _ in range(nsteps)]
EPOCH = int(os.environ.get("LOOP_EPOCH", "240"))   # steps per run_steps call (bench.py: epochs of 12 steps)
L.run_steps(tr, pipe, plan(24))
torch.cuda.synchronize()
stamps = []
t0 = time.perf_counter()
for _ in range(240 // EPOCH):
    L.run_steps(tr, pipe, plan(EPOCH), on_step=lambda s, g, o: stamps.append(time.perf_counter()))
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
d = np.diff(np.array(stamps)) * 1e6
print(f"240 steps: host loop returned after {t_host * 1e3:.1f} ms, device done after {t_all * 1e3:.1f} ms ({t_all / 240 * 1e6:.0f} us per step); "
      f"host per iteration: median {np.median(d):.0f} us, p10 {np.percentile(d, 10):.0f}, p90 {np.percentile(d, 90):.0f}, first 20 median {np.median(d[:20]):.0f}")
