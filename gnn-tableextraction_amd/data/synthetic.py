"""Synthetic PubLayNet-style page graphs: stand-ins for the pre-built graphs the
reference loads from its DGL cache (src/components/graphs/loader.py:115-129) -- real PDFs, PyMuPDF
and the dataset are not available, and BASELINE.json asks for "data": "synthetic".

A page = words laid out in reading order on a 1654 x 2339 canvas; edges = k nearest boxes by the
reference's box distance (src/components/graphs/utils.py:56-88: 0 if the boxes overlap, the axis
gap if they face each other, else int(euclid) between the nearest corners), direction
neighbour -> node (builder.py:226,290), pruned at max_dist (builder.py:287), then
to_simple + to_bidirected (loader.py:313-320); edge weight 1 - d / max d (loader.py:332-344);
13 BBOX features shaped like src/components/nlp/bbox.py:49-107 (9 raw-pixel geometry values +
4-bin character histogram), N(0,1) for the embedding part when F0 > 13; 9 classes, TEXT-dominated.
This is host-side data preparation (numpy); nothing here is on the timed path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

PAGE_W, PAGE_H = 1654, 2339
CLASS_P = np.array([0.62, 0.06, 0.05, 0.04, 0.03, 0.12, 0.04, 0.03, 0.01])


@dataclass
class Page:
    src: np.ndarray       # int32 [E]
    dst: np.ndarray       # int32 [E]
    weight: np.ndarray    # float32 [E]
    feat: np.ndarray      # float32 [n, F0]
    label: np.ndarray     # int64 [n]
    bbox: np.ndarray      # int32 [n, 4]

    @property
    def num_nodes(self) -> int:
        return self.feat.shape[0]


def box_distance_matrix(b: np.ndarray) -> np.ndarray:
    """All-pairs reference box distance (graphs/utils.py:56-88), int64 [n, n]."""
    x0, y0, x1, y1 = (b[:, i].astype(np.int64) for i in range(4))
    dx = np.maximum(np.maximum(x0[None, :] - x1[:, None], x0[:, None] - x1[None, :]), 0)
    dy = np.maximum(np.maximum(y0[None, :] - y1[:, None], y0[:, None] - y1[None, :]), 0)
    diag = (dx > 0) & (dy > 0)
    d = np.maximum(dx, dy)
    d[diag] = np.sqrt((dx[diag] ** 2 + dy[diag] ** 2).astype(np.float64)).astype(np.int64)
    return d


def layout_words(rng: np.random.Generator, n: int) -> np.ndarray:
    """n word boxes in reading order (rows top -> bottom, words left -> right)."""
    boxes = np.zeros((n, 4), dtype=np.int32)
    x, y = 120, 150
    line_h = int(rng.integers(18, 33))
    for i in range(n):
        w = int(rng.integers(15, 121))
        if x + w > PAGE_W - 120:
            x = 120 + int(rng.integers(0, 40))
            y += line_h + int(rng.integers(4, 14))
            line_h = int(rng.integers(18, 33))
            if y + line_h > PAGE_H - 100:
                y = 150            # wrap (dense pages): keeps boxes on the canvas
        boxes[i] = (x, y, x + w, y + line_h)
        x += w + int(rng.integers(6, 18))
    return boxes


def bbox_features(rng: np.random.Generator, b: np.ndarray) -> np.ndarray:
    """13 BBOX features: raw geometry (un-normalised, as in the reference) + 4-bin char histogram."""
    x0, y0, x1, y1 = (b[:, i].astype(np.float32) for i in range(4))
    w, h = x1 - x0, y1 - y0
    geo = np.stack([x0, y0, x1, y1, w, h, w * h, (x0 + x1) / 2, (y0 + y1) / 2], axis=1)
    hist = rng.dirichlet(np.ones(4), size=b.shape[0]).astype(np.float32)
    return np.concatenate([geo, hist], axis=1).astype(np.float32)


def make_page(page_id: int, in_feats: int = 13, n_words: Optional[int] = None, k: int = 5,
              max_dist: int = 500, bidirectional: bool = True, n_classes: int = 9, seed: int = 42) -> Page:
    rng = np.random.default_rng(seed + page_id)
    if n_words is None:
        n_words = int(np.clip(round(rng.lognormal(np.log(200.0), 0.6)), 20, 2000))
    b = layout_words(rng, n_words)
    d = box_distance_matrix(b)
    np.fill_diagonal(d, np.iinfo(np.int64).max)
    kk = min(k, n_words - 1)
    nbr = np.argpartition(d, kk - 1, axis=1)[:, :kk] if kk > 0 else np.zeros((n_words, 0), dtype=np.int64)
    node = np.repeat(np.arange(n_words), kk)
    nb = nbr.reshape(-1)
    keep = d[node, nb] <= max_dist
    src, dst = nb[keep], node[keep]                       # neighbour -> node
    if bidirectional:                                     # to_simple + to_bidirected
        key = np.unique(np.concatenate([src * n_words + dst, dst * n_words + src]))
        src, dst = key // n_words, key % n_words
    else:
        key = np.unique(src * n_words + dst)
        src, dst = key // n_words, key % n_words
    dist = d[dst, src].astype(np.float64)
    m = dist.max() if dist.size and dist.max() > 0 else 1.0
    weight = (1.0 - dist / m).astype(np.float32)
    feat = bbox_features(rng, b)
    if in_feats > 13:
        feat = np.concatenate([feat, rng.standard_normal((n_words, in_feats - 13)).astype(np.float32)], axis=1)
    elif in_feats < 13:
        feat = feat[:, :in_feats]
    p = CLASS_P[:n_classes] / CLASS_P[:n_classes].sum()
    label = rng.choice(n_classes, size=n_words, p=p).astype(np.int64)
    return Page(src.astype(np.int32), dst.astype(np.int32), weight, np.ascontiguousarray(feat), label, b)


def make_pages(n_pages: int, in_feats: int = 13, first_id: int = 0, **kw) -> List[Page]:
    return [make_page(first_id + i, in_feats=in_feats, **kw) for i in range(n_pages)]


def concat_pages(pages: List[Page]):
    """Block-diagonal union on the host (what dgl.batch does): returns src, dst, weight, feat, label,
    node offsets."""
    off = np.zeros(len(pages) + 1, dtype=np.int64)
    for i, p in enumerate(pages):
        off[i + 1] = off[i] + p.num_nodes
    src = np.concatenate([p.src.astype(np.int64) + off[i] for i, p in enumerate(pages)]).astype(np.int32)
    dst = np.concatenate([p.dst.astype(np.int64) + off[i] for i, p in enumerate(pages)]).astype(np.int32)
    weight = np.concatenate([p.weight for p in pages])
    feat = np.concatenate([p.feat for p in pages], axis=0)
    label = np.concatenate([p.label for p in pages])
    return src, dst, weight, feat, label, off


def make_knn_stress_graph(n: int = 1_000_000, k: int = 12, seed: int = 42):
    """BASELINE cfg4: one graph of n 2-D uniform points in Morton (Z-curve) order, k nearest
    neighbours each (E = n*k, in-degree exactly k), w ~ U(0,1).  Returns src, dst, weight."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    pts = rng.random((n, 2))
    q = (pts * 65535).astype(np.uint64)

    def spread(v):
        v = (v | (v << 8)) & np.uint64(0x00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x33333333)
        v = (v | (v << 1)) & np.uint64(0x55555555)
        return v

    order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)), kind="stable")
    pts = pts[order]
    _, nbr = cKDTree(pts).query(pts, k=k + 1, workers=-1)
    src = nbr[:, 1:].reshape(-1).astype(np.int32)
    dst = np.repeat(np.arange(n, dtype=np.int32), k)
    weight = rng.random(n * k).astype(np.float32)
    return src, dst, weight
