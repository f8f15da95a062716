"""A training set several times the HBM budget through the windowed residency (models/residency.py), at a size where the 4 GB
row-map bound -- not the budget -- sets the window:  python profiles/residency_scale.py [pages=16000] [budget_GB=8] [passes=8] [steps=600]
Prints one JSON object: set size, windows, steady-state nodes/s of the windowed loop, upload rate, the all-resident rate of the
same step stream on a subset that fits, host build time."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop, residency as R
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
import bench

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
budget = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else 8e9
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 600
F0, B = 831, 100
t0 = time.perf_counter()
pages = bench.make_pages_parallel(n_pages, F0, 0, min(32, os.cpu_count() or 1))      # (before anything initialises the GPU: fork)
gen_s = time.perf_counter() - t0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
graphs = bench.to_page_graphs(gte, pages)

def fresh():
    torch.manual_seed(42)
    m = gte.GcnSAGE(F0, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
    return FusedGcnSageStep(m, lr=0.01, weight_decay=5e-4)
tr = fresh()
want_p3 = bool(tr.wants_resident_images(F0))        # (what train(), run_steps and predict_resident decide with)
t0 = time.perf_counter()
host = R.HostPages(graphs, dev)
build_s = time.perf_counter() - t0
want_agg = bool(want_p3 and tr.wants_agg_image(F0))
per_node = R.WindowedPages.bytes_per_node(host.page_nodes, host.page_edges, F0, want_p3, want_agg)
set_bytes = float(host.page_nodes.sum()) * per_node
wp = R.WindowedPages(host, budget, want_p3, want_agg)
stream = R.WindowStream(wp.ranges, B, passes, 42)
wp.prefetch(stream.peek_window())
pipe = loop.BatchPipeline(wp.acquire(stream.peek_window()))
pipe._bound_pages = (host.page_nodes, np.diff(host.sets["in"]["edge_off"]), np.diff(host.sets["out"]["edge_off"]))
R.run_windowed(tr, pipe, wp, stream, 48)
torch.cuda.synchronize()
up0, nodes = wp.uploaded_bytes, [0]
t0 = time.perf_counter()
out3, _ = R.run_windowed(tr, pipe, wp, stream, n_steps, on_step=lambda s, g, o: nodes.__setitem__(0, nodes[0] + g.num_nodes()))
torch.cuda.synchronize()
el = time.perf_counter() - t0
res = {"pages": n_pages, "nodes": int(host.page_nodes.sum()), "set_GB_resident_form": set_bytes / 1e9, "pinned_host_GB": host.feature_bytes() / 1e9,
       "budget_GB": budget / 1e9, "device_GB": wp.device_bytes / 1e9, "windows": len(wp.ranges),
       "pages_per_window": [p1 - p0 for p0, p1 in wp.ranges][:4], "passes": passes, "steps": n_steps,
       "windowed_nodes_per_s": nodes[0] / el, "ms_per_step": el / n_steps * 1e3, "upload_GB_per_s": (wp.uploaded_bytes - up0) / el / 1e9,
       "final_loss": float(out3[0]), "page_generation_s": gen_s, "host_build_s": build_s}
del pipe, tr
# the all-resident rate: the same kind of step stream on the first window's pages, resident
p0, p1 = wp.ranges[0]
tr2 = fresh()
rp = G.ResidentPages(graphs[p0:p1], dev)
if want_p3:
    rp.enable_p3(agg=want_agg)
pipe2 = loop.BatchPipeline(rp)
rng = np.random.default_rng(0)
plan = lambda k: [np.sort(rng.choice(p1 - p0, B, replace=False)) for _ in range(k)]
loop.run_steps(tr2, pipe2, plan(24))
torch.cuda.synchronize()
t0 = time.perf_counter()
pl = plan(240)
loop.run_steps(tr2, pipe2, pl)
torch.cuda.synchronize()
el2 = time.perf_counter() - t0
n2 = sum(pipe2.nodes(i) for i in range(240))
res["all_resident_nodes_per_s"] = n2 / el2
res["windowed_over_all_resident"] = res["windowed_nodes_per_s"] / res["all_resident_nodes_per_s"]
print(json.dumps(res))
