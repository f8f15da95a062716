"""The validation forward on one giant graph (all pages of a set in one graph, SURVEY 8(d) cfg2 'val graph'): the module path
model(g) against the step engine's forward-only call (engine.forward_logits -> gte_gcnsage_forward)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
dev = "cuda:0"
for n_pages in (100, 500, 2000):
    pages = S.make_pages(n_pages, in_feats=831)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    g = gte.PageGraph(src, dst, int(off[-1]), device=dev)
    g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
    torch.manual_seed(0)
    model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0).to(dev).eval()
    tr = FusedGcnSageStep(model)
    def t(fn, reps=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    with torch.no_grad():
        a = t(lambda: model(g))
        b = t(lambda: tr.forward_logits(g))
        d = float((model(g) - tr.forward_logits(g)).abs().max())
    print(f"{n_pages} pages, {int(off[-1])} nodes: module {a:.3f} ms, engine forward {b:.3f} ms, max |diff| {d:.2e}")
    del tr, model, g
    torch.cuda.empty_cache()
