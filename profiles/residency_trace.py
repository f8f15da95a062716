"""Where the time of a host-resident (windowed) training run goes: host time per chunk section and device-side waits.
    python profiles/residency_trace.py [pages=1200] [passes=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop, residency as R
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.6
dev = torch.device("cuda", 0)
pages = S.make_pages(n_pages, in_feats=831)
graphs = []
for p in pages:
    g = gte.PageGraph(p.src, p.dst, p.num_nodes)
    g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
    g.edata["feat"] = torch.from_numpy(p.weight)
    graphs.append(g)
torch.manual_seed(0)
model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
host = R.HostPages(graphs, dev)
per_node = R.WindowedPages.bytes_per_node(host.page_nodes, host.page_edges, 831, True)
wp = R.WindowedPages(host, float(host.page_nodes.sum()) * per_node / frac, True)
print("windows:", [(b - a) for a, b in wp.ranges])
stream = R.WindowStream(wp.ranges, 100, passes, 42)
wp.prefetch(stream.peek_window())
pipe = loop.BatchPipeline(wp.acquire(stream.peek_window()))
pipe._bound_pages = (host.page_nodes, np.diff(host.sets["in"]["edge_off"]), np.diff(host.sets["out"]["edge_off"]))
R.run_windowed(tr, pipe, wp, stream, 32)
torch.cuda.synchronize()
T = {"acquire": 0.0, "rebind": 0.0, "prefetch": 0.0, "run_steps": 0.0, "release": 0.0}
n_steps, done = 160, 0
t_all = time.perf_counter()
chunks = stream.take(n_steps)
wait_ms = 0.0
for i, (w, steps) in enumerate(chunks):
    t0 = time.perf_counter(); res = wp.acquire(w); T["acquire"] += time.perf_counter() - t0
    sl = wp._slot_of(w)
    ready_done = sl["ready"].query()
    t0 = time.perf_counter()
    if pipe.res is not res:
        pipe.rebind(res)
    T["rebind"] += time.perf_counter() - t0
    nxt = chunks[i + 1][0] if i + 1 < len(chunks) else (stream.peek_window() if stream.peek_window() != w else stream.next_window())
    t0 = time.perf_counter()
    if nxt != w:
        wp.prefetch(nxt)
    T["prefetch"] += time.perf_counter() - t0
    t0 = time.perf_counter(); loop.run_steps(tr, pipe, steps); T["run_steps"] += time.perf_counter() - t0
    t0 = time.perf_counter(); wp.release(w); T["release"] += time.perf_counter() - t0
    print(f"chunk {i}: window {w} steps {len(steps)} upload-done-at-acquire {ready_done}")
torch.cuda.synchronize()
el = time.perf_counter() - t_all
print(f"{n_steps} steps in {el * 1e3:.1f} ms = {el / n_steps * 1e3:.3f} ms/step; host sections (ms):", {k: round(v * 1e3, 1) for k, v in T.items()})

# host time of the statements inside prefetch(): wrap the pieces
import types
_orig_copy = torch.Tensor.copy_
tc = {"copy_ms": 0.0, "n": 0, "max_ms": 0.0}
def timed_copy(self, src, non_blocking=False):
    t0 = time.perf_counter(); r = _orig_copy(self, src, non_blocking=non_blocking); dt = (time.perf_counter() - t0) * 1e3
    tc["copy_ms"] += dt; tc["n"] += 1; tc["max_ms"] = max(tc["max_ms"], dt)
    if dt > 0.5: print(f"   slow copy_: {dt:.2f} ms, {self.numel() * self.element_size() / 1e6:.1f} MB, src pinned={src.is_pinned() if not src.is_cuda else None} dtype {src.dtype}")
    return r
torch.Tensor.copy_ = timed_copy
t0 = time.perf_counter()
R.run_windowed(tr, pipe, wp, stream, 96)
print("host time of 96 windowed steps:", (time.perf_counter() - t0) * 1e3, "ms; copy_ calls:", tc)
torch.Tensor.copy_ = _orig_copy
torch.cuda.synchronize()
# host profile of the windowed loop
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
R.run_windowed(tr, pipe, wp, stream, 96)
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28)
print(st.getvalue()[:6000])
