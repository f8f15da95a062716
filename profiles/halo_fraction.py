"""Kill criterion of the "aggregation in the GEMM's prologue" fusion (DESIGN "tried and killed"): a hidden layer's
GEMM could aggregate its own A tile in LDS if (almost) every edge's SOURCE row lay inside the destination's row tile plus a small
halo.  This script measures that on the synthetic PubLayNet-style pages the bench trains on (data/synthetic.py: k-NN of word boxes,
bidirected, pages concatenated): the fraction of edges whose source lies within the destination's 128-row (96-row) tile +- 64 rows.
CPU only:  python profiles/halo_fraction.py  ->  profiles/r06/halo_fraction.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gnn_tableextraction_amd.data import synthetic as S  # noqa: E402

pages = S.make_pages(100, in_feats=13)
src, dst, w, feat, label, off = S.concat_pages(pages)
n = int(off[-1])
out = [f"100 synthetic pages, {n} nodes, {len(src)} edges (mean in-degree {len(src) / n:.2f}); row order = the order the batch is assembled in"]
for tile in (128, 96):
    t0 = (dst // tile) * tile
    for halo in (0, 32, 64, 128):
        inside = (src >= t0 - halo) & (src < t0 + tile + halo)
        out.append(f"tile {tile:3d} rows, halo +-{halo:3d}: {inside.mean() * 100:6.2f} % of the edges have their source inside "
                   f"(LDS for the A tile + halo at 256 fp32 columns: {(tile + 2 * halo) * 1024 / 1024:.0f} KB)")
d = np.abs(src.astype(np.int64) - dst.astype(np.int64))
out.append(f"|src - dst| in rows: median {np.median(d):.0f}, 90 % {np.percentile(d, 90):.0f}, 99 % {np.percentile(d, 99):.0f}, max {d.max()} "
           f"(a page has ~{n // 100} rows; words are ordered by reading order, neighbours in the k-NN graph lie lines apart)")
print("\n".join(out))
