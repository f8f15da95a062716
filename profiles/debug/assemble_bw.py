"""Batch assembly (gte_batch_assemble) alone: GB/s by feature width (831: rows 4-byte aligned only; 832: 16-byte aligned)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_tableextraction_amd import graph as G, distributed as D
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models import loop
dev = torch.device("cuda", 0)
for f0 in (831, 832, 13, 256):
    pages = S.make_pages(400, in_feats=f0)
    gs = []
    for p in pages:
        g = G.PageGraph(p.src, p.dst, p.num_nodes)
        g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
        g.edata["feat"] = torch.from_numpy(p.weight)
        gs.append(g)
    res = G.ResidentPages(gs, dev)
    plan = [r[0] for r in D.plan_epoch(res.page_sizes(), 100, 1, seed=1, epoch=0)]
    pipe = loop.BatchPipeline(res, side_stream=False)
    pipe.load(plan)
    for s in range(len(plan)):
        pipe.start(s); pipe.get(s); pipe.release(s)
    torch.cuda.synchronize()
    evs = []
    for rep in range(5):
        for s in range(len(plan)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); pipe.start(s); b.record(); pipe.get(s); pipe.release(s)
            evs.append((a, b, pipe.nodes(s)))
    torch.cuda.synchronize()
    us = np.mean([a.elapsed_time(b) for a, b, _ in evs]) * 1e3
    n = np.mean([k for _, _, k in evs])
    print(f"F0={f0}: assembly {us:.1f} us per batch of {n:.0f} nodes; features {2 * n * f0 * 4 / us / 1e3:.0f} GB/s (read + write)")
