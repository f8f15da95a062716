#!/bin/bash
# kernels of the forward-only paths on a 100-page graph: module path vs engine.forward_logits
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/vf; mkdir -p $O
cat > /tmp/vf.py <<PY
import sys, os
sys.path.insert(0, "$R")
import torch, gnn_tableextraction_amd as gte
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
dev = "cuda:0"
pages = S.make_pages(100, in_feats=831)
src, dst, w, feat, label, off = S.concat_pages(pages)
g = gte.PageGraph(src, dst, int(off[-1]), device=dev)
g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
torch.manual_seed(0)
model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0).to(dev).eval()
tr = FusedGcnSageStep(model)
mode = sys.argv[1]
with torch.no_grad():
    for _ in range(50):
        (model(g) if mode == "module" else tr.forward_logits(g))
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for m in module engine; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/t_$m -o t -- python3 /tmp/vf.py $m > $O/$m.log 2>&1
  python3 $R/profiles/rocpd_summary.py $(ls $O/t_$m/*.db | head -1) $O/stats_$m.csv > /dev/null
  rm -rf $O/t_$m
  echo "== $m"; head -12 $O/stats_$m.csv | cut -c1-120
done
