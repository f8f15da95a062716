"""The validation-graph forward (all pages of a validation set in ONE graph, forward only: model_train.py:246,349-353 of the
reference) through engine.forward_logits on the cached feature image.  For rocprofv3 --kernel-trace:
    python profiles/val_forward.py [pages=2000] [in_feats=831] [hidden=256] [reps=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
f0 = int(sys.argv[2]) if len(sys.argv) > 2 else 831
hid = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda", 0)
pages = S.make_pages(n_pages, in_feats=f0)
src, dst, w, feat, label, off = S.concat_pages(pages)
n = int(off[-1])
g = gte.PageGraph(src, dst, n, device=dev)
g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
torch.manual_seed(0)
model = gte.GcnSAGE(f0, hid, 9, 3, torch.nn.functional.relu, 0).to(dev)
eng = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
print("image:", eng.attach_feature_image(g), "kinds:", eng._plan_kinds(f0, n, eng._batch_cached(g)))
for _ in range(3):
    eng.forward_logits(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.forward_logits(g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{n} nodes: {dt * 1e3:.3f} ms per forward = {n / dt / 1e6:.1f} M nodes/s")
