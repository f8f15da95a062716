"""Where the time of the train loop goes (MI355X): the same steps as
  replay      HIP-graph replay of 4 captured resident batches                 (device-time floor of a step)
  eager       eager launches of the same 4 resident batches                   (+ dispatch gaps / host launch rate)
  loop_main   models/loop.py with the batch assembly on the step's own stream (+ assembly in front of every step)
  loop_side   models/loop.py as shipped: assembly one step ahead on a side stream
For every mode: ms/step, and the host time needed to QUEUE a step (if that exceeds the device time the loop is host-bound).
usage: python profiles/loop_bench.py [in_feats] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G, distributed as D
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
f0 = int(sys.argv[1]) if len(sys.argv) > 1 else 831
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda", 0)
pages = S.make_pages(600, in_feats=f0)
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
plan = [ids for ep in range(40) for ids in (r[0] for r in D.plan_epoch(sizes, 100, 1, seed=42, epoch=ep))]
fixed = [res.batch(ids) for ids in plan[:4]]


def report(name, fn, n):
    fn(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nodes = fn(n)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"F0={f0} {name:10s}: {dt/n*1e3:.3f} ms/step  {nodes/dt/1e6:.2f} M nodes/s   host queueing {t_host/n*1e3:.3f} ms/step", flush=True)


replays = [step.capture(g, g.ndata["label"]) for g in fixed]
def run_replay(n):
    for i in range(n): replays[i % 4]()
    return sum(fixed[i % 4].num_nodes() for i in range(n))
def run_eager(n):
    for i in range(n): step.step(fixed[i % 4], fixed[i % 4].ndata["label"])
    return sum(fixed[i % 4].num_nodes() for i in range(n))
def run_loop(pipe):
    def f(n):
        tot, k = 0, 0
        while k < n:                                   # epochs of 6 steps, as train() would run them
            chunk = plan[k % 200: k % 200 + min(6, n - k)]
            loop.run_steps(step, pipe, chunk)
            tot += sum(pipe.nodes(i) for i in range(len(chunk)))
            k += len(chunk)
        return tot
    return f
def run_long(pipe):
    def f(n):
        loop.run_steps(step, pipe, plan[:n])
        return sum(pipe.nodes(i) for i in range(n))
    return f
pm, ps = loop.BatchPipeline(res, side_stream=False), loop.BatchPipeline(res, side_stream=True)
for rep in range(int(os.environ.get("REPS", "3"))):
    if not os.environ.get("ONLY_LOOP"):
        report("replay", run_replay, steps)
        report("eager", run_eager, steps)
        report("main/6", run_loop(pm), steps)
        report("side/6", run_loop(ps), steps)
    report("main/1ep", run_long(pm), steps)
    report("side/1ep", run_long(ps), steps)
