import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gnn_tableextraction_amd import ops
from gnn_tableextraction_amd.data import synthetic as S
dev="cuda:0"
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps*1e3
for npages in (200, 400, 800, 1600):
    pages=S.make_pages(npages, in_feats=13)
    src,dst,w,feat,label,off=S.concat_pages(pages); n=int(off[-1])
    ip,ix,perm,wt=ops.coo_to_csr(torch.from_numpy(dst).to(dev), torch.from_numpy(src).to(dev), n, torch.from_numpy(w).to(dev))
    plan=ops.build_tile_plan(ip,ix,n)
    for f in (256, 512, 831):
        x=torch.randn(n,f,device=dev); out=torch.empty_like(x)
        a=timeit(lambda: ops.spmm_csr(ip,ix,wt,x,n,mean=True,out=out))
        b=timeit(lambda: ops.spmm_csr(ip,ix,wt,x,n,mean=True,out=out,tiles=plan,force_tiled=True))
        print(f"pages {npages:5d} n={n:7d} F={f:4d} ({n*f*4/1e6:7.1f} MB): plain {a:8.1f} us  tiled {b:8.1f} us  {'TILED' if b<a else 'plain'}", flush=True)
