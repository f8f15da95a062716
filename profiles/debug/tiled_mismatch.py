"""Where does the LDS-staged aggregation differ from the row-per-wave kernel?  (debug aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_tableextraction_amd import ops
from gnn_tableextraction_amd.data import synthetic as S
dev = "cuda:0"
rng = np.random.default_rng(1002)
pages = S.make_pages(40, in_feats=13, first_id=77)
src, dst, w, feat, label, off = S.concat_pages(pages)
n = int(off[-1])
ip, ix, perm, wt = ops.coo_to_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, torch.from_numpy(w).to(dev))
plan = ops.build_tile_plan(ip, ix, n)
tp = plan.tile_ptr.cpu().numpy(); ipn = ip.cpu().numpy()
nt = len(tp) - 1
print("n", n, "tiles", nt, "nu", [int(tp[t + 1] - tp[t]) for t in range(nt)])
print("ne", [int(ipn[min((t + 1) * 32, n)] - ipn[t * 32]) for t in range(nt)])
for f, pad in ((352, 4), (448, 8), (512, 0), (384, 0)):
    xs = torch.randn(n, f + pad, device=dev); x = xs[:, :f]
    a = ops.spmm_csr(ip, ix, wt, x, n, mean=True)
    outs = torch.zeros(n, f + pad, device=dev)
    for rep in range(8):
        b = ops.spmm_csr(ip, ix, wt, x, n, mean=True, out=outs[:, :f], tiles=plan, force_tiled=True)
        bad = (a != b).nonzero()
        if len(bad):
            rows = bad[:, 0].unique().tolist(); cols = bad[:, 1].unique().tolist()
            print(f"f={f} pad={pad} rep={rep}: {len(bad)} bad, tiles {sorted(set(r // 32 for r in rows))} cols {min(cols)}..{max(cols)}")
        else:
            print(f"f={f} pad={pad} rep={rep}: equal")
