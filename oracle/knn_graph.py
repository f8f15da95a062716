"""CPU oracle for the k-NN page-graph construction  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, in numpy, what the reference does between word boxes and the graph the model sees:

  builder.py:383-395   pixel projections: node j owns the x pixels range(x0, x1) and the y pixels range(y0, y1)
  builder.py:240-292   knn(): grow a window around the node (multiplier 2, 3, ... 99; a box wider than tall grows
                       4x faster vertically than horizontally and vice versa) until it holds >= k boxes (the node itself
                       counts) ; take the k nearest of the LAST window by ``distance`` (graphs/utils.py:56-88) ; keep those
                       within max_dist ; direction neighbour -> node ; an edge whose reverse was added while an EARLIER
                       node was processed is skipped
  loader.py:313-320    dgl.to_simple + dgl.to_bidirected
  builder.py:567-582   fast_remove_islands: TEXT nodes with no non-TEXT node at the end of any walk of exactly
                       ``range_island`` steps (dgl.khop_adj = A^k) on the bidirected graph
  loader.py:332-344    edge weights (oracle/box_geometry.py)

Determinism.  The reference orders equal-distance candidates by CPython ``set`` iteration order followed by numpy's
default (unstable) argsort, i.e. by an implementation detail.  This restatement breaks ties by (distance, node id).  The
two agree exactly wherever the choice is unique: a node is *ambiguous* iff its k-th and (k+1)-th candidate distances are
equal (ties INSIDE the first k do not change the selected set).  ``tests/test_knn_graph.py`` pins the restatement on the
reference's own output (fixture written by oracle/make_aux_golden.py from the ast-extracted ``get_edges``): identical
edges at every unambiguous node, identical selected DISTANCES at ambiguous ones.

Boxes that leave the canvas are kept the way the reference's per-pixel lists keep them (``projection_hits``): pixels past
the right / bottom edge count for the last column / row, negative pixels wrap around (Python indexing; below -extent the
reference raises IndexError -- not reproduced).
"""
import numpy as np

from . import box_geometry as bg

TEXT = 1          # Categories_names.TEXT.value (src/utils/const.py:6)


def window_of(b, m, width, height):
    """The search window of multiplier m (builder.py:256-267).  Python int() truncates toward zero; all operands >= 0."""
    w, h = int(b[2] - b[0]), int(b[3] - b[1])
    if w > h:
        ho, vo = int(w * m / 4), int(h * m)
    else:
        ho, vo = int(w * m), int(h * m / 4)
    lo = lambda a: 0 if a < 0 else a
    x0, y0 = lo(int(b[0]) - ho), lo(int(b[1]) - vo)
    x1 = lo(int(b[2]) + ho)
    y1 = lo(int(b[3]) + vo)
    x1 = width if x1 > width else x1
    y1 = height if y1 > height else y1
    return x0, y0, x1, y1


def candidates(bboxs, i, m, width, height):
    """ids of the boxes with a pixel column in [x0, x1) and a pixel row in [y0, y1) of node i's window (builder.py:266-271)"""
    b = np.asarray(bboxs, dtype=np.int64)
    x0, y0, x1, y1 = window_of(b[i], m, width, height)
    ok = projection_hits(b[:, 0], b[:, 2], x0, x1, width) & projection_hits(b[:, 1], b[:, 3], y0, y1, height)
    return np.nonzero(ok)[0]


def projection_hits(lo, hi, w0, w1, extent):
    """Does a box with pixels range(lo, hi) own a projection slot inside [w0, w1)?  builder.py:386-394 files pixel hp under
    slot hp, pixels >= extent under the LAST slot (``if hp >= width: hp = width - 1``) and -- Python indexing -- a negative
    pixel hp under slot extent + hp.  Boxes wholly on the canvas: the plain interval overlap."""
    lo, hi = np.asarray(lo, dtype=np.int64), np.asarray(hi, dtype=np.int64)
    inside = np.maximum(np.maximum(lo, 0), w0) < np.minimum(np.minimum(hi, extent), w1)
    over = (hi > np.maximum(lo, extent)) & (w0 <= extent - 1) & (extent - 1 < w1)
    neg = (lo < 0) & (np.maximum(lo + extent, w0) < np.minimum(np.minimum(hi, 0) + extent, w1))
    return inside | over | neg


def knn_select(bboxs, size, k, max_dist):
    """Per node: (selected neighbour ids ordered by (distance, id), their distances, ambiguous flag)."""
    b = np.asarray(bboxs, dtype=np.int64)
    width, height = int(size[0]), int(size[1])
    out = []
    for i in range(len(b)):
        cand = np.zeros(0, dtype=np.int64)
        m = 2
        while len(cand) < k and m < 100:
            cand = candidates(b, i, m, width, height)
            m += 1
        cand = cand[cand != i]
        d = bg.distance_many(b[i], b[cand])
        order = np.lexsort((cand, d))                       # by distance, then node id
        sel, ds = cand[order][:k], d[order][:k]
        ambiguous = len(cand) > k and d[order][k] == d[order][k - 1] and ds[-1] <= max_dist
        keep = ds <= max_dist
        out.append((sel[keep], ds[keep], bool(ambiguous)))
    return out


def knn_edges(bboxs, size, k=5, max_dist=500):
    """(u, v) of builder.py:240-292 in the reference's append order, with the deterministic tie-break."""
    sel = knn_select(bboxs, size, k, max_dist)
    accepted = [set(s[0].tolist()) for s in sel]
    u, v = [], []
    for i, (nbrs, _, _) in enumerate(sel):
        for n in nbrs.tolist():
            # `[node_index, neighbor] not in edges`: the edge node_index -> neighbor exists iff neighbor was processed
            # before (neighbor < node_index), selected node_index, and that edge was not skipped itself (it cannot have been)
            if n < i and i in accepted[n]:
                continue
            u.append(n)
            v.append(i)
    return np.asarray(u, dtype=np.int64), np.asarray(v, dtype=np.int64)


def to_simple_bidirected(u, v, n):
    """dgl.to_simple + dgl.to_bidirected (loader.py:319-320): the symmetric closure without duplicates, sorted by
    (dst, src) -- the in-edge CSR order this repository uses."""
    key = np.unique(np.concatenate([u * n + v, v * n + u]))
    src, dst = key // n, key % n
    order = np.lexsort((src, dst))
    return src[order], dst[order]


def island_nodes(src, dst, labels, n, khop=2, text=TEXT):
    """fast_remove_islands (builder.py:567-582) on an already simple + bidirected edge list: TEXT nodes from which no walk
    of exactly ``khop`` steps ends at a non-TEXT node."""
    labels = np.asarray(labels)
    reach = (labels != text)
    for _ in range(khop):
        nxt = np.zeros(n, dtype=bool)
        np.logical_or.at(nxt, dst, reach[src])               # node dst has a neighbour src from which ...
        reach = nxt
    return np.nonzero((~reach) & (labels == text))[0]


# ---- reference side: build container only -----------------------------------------------------------------------------
def reference_get_edges(bboxs, size, k, max_dist, mode="knn"):
    """Runs the reference's nested ``get_edges`` (builder.py:222-411), cut out of its file with ``ast``.  Needs
    /root/reference: used by oracle/make_aux_golden.py only."""
    import torch
    from .make_aux_golden import extract
    from math import inf, sqrt
    distance = extract("components/graphs/utils.py", "distance", glb={"sqrt": sqrt, "inf": inf})
    cfg = type("C", (), {})()
    cfg.PREPROCESS = type("P", (), {"k": k, "max_dist": max_dist})()
    me = type("B", (), {"config": cfg})()
    glb = {"np": np, "torch": torch, "distance": distance, "bboxs": [list(map(int, b)) for b in bboxs],
           "size": (int(size[0]), int(size[1])), "self": me}
    get_edges = extract("components/graphs/builder.py", "get_edges", glb=glb)
    u, v = get_edges(mode)
    return u.numpy().astype(np.int64), v.numpy().astype(np.int64)


def fixture_pages(seed=7):
    """Seeded word layouts: reading-order lines of words, some dense, some sparse, some with overlapping boxes."""
    rng = np.random.default_rng(seed)
    pages = []
    for p in range(10):
        width, height = int(rng.integers(500, 900)), int(rng.integers(600, 1100))
        n = int(rng.integers(12, 90))
        boxes, x, y = [], 30, 40
        lh = int(rng.integers(9, 18))
        for _ in range(n):
            w = int(rng.integers(8, 70))
            if x + w > width - 30:
                x = 30 + int(rng.integers(0, 25))
                y += lh + int(rng.integers(2, 30))
                lh = int(rng.integers(9, 18))
            if y + lh > height - 20:
                y = 40 + int(rng.integers(0, 9))
            boxes.append([x, y, x + w, y + lh])
            x += w + int(rng.integers(3, 40))
        b = np.asarray(boxes, dtype=np.int64)
        if p % 3 == 2:                                       # a few overlapping / zero-size boxes
            b[1] = b[0]
            b[5, 2] = b[5, 0]
        pages.append((b, (width, height), int(rng.choice([3, 5, 5, 8])), int(rng.choice([60, 500, 500]))))
    return pages


def big_fixture_pages(seed=23, sizes=(300, 700, 1500, 3000)):
    """Pages at the workload's real sizes (LogNormal(200) words up to 2 000; here up to 3 000) on an A4 canvas
    at 1/SCALE_FACTOR (1654 x 2339): the device kernels then run several workgroups per page (256 nodes each) and stage
    thousands of boxes in LDS.  Word sizes shrink with the word count so the page fills without wrapping around."""
    rng = np.random.default_rng(seed)
    pages = []
    for p, n in enumerate(sizes):
        width, height = 1654, 2339
        dense = n > 1000
        boxes, x, y = [], 30, 40
        lh = int(rng.integers(7, 11)) if dense else int(rng.integers(10, 22))
        for _ in range(n):
            w = int(rng.integers(6, 40)) if dense else int(rng.integers(12, 90))
            if x + w > width - 30:
                x = 30 + int(rng.integers(0, 25))
                y += lh + (int(rng.integers(1, 6)) if dense else int(rng.integers(2, 40)))
                lh = int(rng.integers(7, 11)) if dense else int(rng.integers(10, 22))
            if y + lh > height - 20:
                y = 40 + int(rng.integers(0, 9))                       # wrap: overlapping lines (scanned two-column pages do this)
            boxes.append([x, y, x + w, y + lh])
            x += w + (int(rng.integers(2, 14)) if dense else int(rng.integers(3, 40)))
        b = np.asarray(boxes, dtype=np.int64)
        if p % 2 == 1:
            b[1] = b[0]
            b[7, 2] = b[7, 0]
        pages.append((b, (width, height), 5, 500))
    return pages


def out_of_canvas_page(seed=31):
    """OCR boxes that leave the page: past the right and the bottom edge (clamped into the last projection slot by the
    reference) and slightly negative (Python's negative indexing wraps them to the far side)."""
    b, size, k, md = fixture_pages(seed)[3]
    b = b.copy()
    w, h = size
    b[2, 2] = w + 25                      # sticks out to the right
    b[4, [0, 2]] = [w + 5, w + 40]        # entirely to the right of the canvas
    b[6, 3] = h + 12
    b[9, [1, 3]] = [h + 3, h + 20]        # entirely below
    b[0, 0] = -4                          # starts left of the canvas
    b[11, 1] = -2
    return (b, size, 5, 500)


def write_reference_fixture(path):
    out = {}
    pages = fixture_pages() + big_fixture_pages() + [out_of_canvas_page()]
    for i, (b, size, k, max_dist) in enumerate(pages):
        u, v = reference_get_edges(b, size, k, max_dist)
        out[f"bbox{i}"], out[f"size{i}"] = b.astype(np.int32), np.asarray(size, dtype=np.int32)
        out[f"k{i}"], out[f"maxd{i}"] = np.int32(k), np.int32(max_dist)
        out[f"u{i}"], out[f"v{i}"] = u.astype(np.int32), v.astype(np.int32)
    out["n_pages"] = np.int32(len(pages))
    np.savez_compressed(path, **out)
