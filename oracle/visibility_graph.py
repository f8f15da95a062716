"""CPU oracle for the VISIBILITY page-graph construction  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates ``GraphBuilder.get_graph(..., mode='visibility')`` between word boxes and the directed edge list:

  builder.py:294-348   visibility(): every node scans all other boxes IN INDEX ORDER and keeps one neighbour per direction
                       (top 0, right 1, bottom 2, left 3; direction = comparison of the box centres):
                         * a box that intersects the node's box (closed intervals) becomes its top / bottom neighbour at
                           distance 0, unconditionally (the LAST such box wins);
                         * a box that overlaps only in x is a top candidate at distance node.y0 - other.y1 if
                           height / 2 > current distance > candidate distance, else a bottom candidate at distance
                           other.y0 - node.y1 if current distance > candidate distance (no height test there);
                         * a box that overlaps only in y likewise right (with the width / 2 test) / left (without);
                         * the current distance starts at PREPROCESS.max_dist;
                       then appends  [top, node] unless [node, top] is already a vertical edge,  [node, right] unless
                       [right, node] is already a horizontal edge,  [node, bottom] unless [bottom, node] is vertical,
                       [left, node] unless [node, left] is horizontal -- in that order, per node, nodes in index order;
  builder.py:350-379   remove_vertical(): a vertical edge is dropped when its centre-to-centre segment crosses the segment of
                       any horizontal edge whose END point differs from the vertical edge's START point (the reference tests
                       ``v1 != h2`` twice); result = remaining vertical edges, then the horizontal edges.

All comparisons are between integers and half-integers: restated on doubled integer coordinates (exact).  Pinned on the
reference's own output: the ast-extracted ``get_edges('visibility')`` run by oracle/make_aux_golden.py on seeded pages
(tests/golden/aux_visibility_edges.npz), identical (u, v) INCLUDING order and duplicates (tests/test_knn_graph.py).
"""
import numpy as np


def centres2(b):
    """doubled box centres (x0 + x2, y1 + y3): builder.py:300 ``x2 - (x2 - x0) / 2`` times two"""
    b = np.asarray(b, dtype=np.int64)
    return np.stack([b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]], 1)


def visibility_select(bboxs, size, max_dist):
    """[n, 4] neighbour ids (top, right, bottom, left; -1 = none) and their distances.
    The reference scans, per node i, every other box j IN INDEX ORDER with order-dependent update rules; here the scan over j
    stays sequential and the nodes i are a numpy vector (a 3 000-word page: seconds instead of minutes)."""
    b = np.asarray(bboxs, dtype=np.int64).reshape(-1, 4)
    n = len(b)
    width, height = int(size[0]), int(size[1])
    c2 = centres2(b) if n else np.zeros((0, 2), dtype=np.int64)
    ids = np.arange(n)
    nb = np.tile(ids[:, None], (1, 4))
    cur = np.full((n, 4), int(max_dist), dtype=np.int64)
    for j in range(n):
        other = ids != j
        top, bottom = c2[j, 1] < c2[:, 1], c2[:, 1] < c2[j, 1]
        right, left = c2[:, 0] < c2[j, 0], c2[j, 0] < c2[:, 0]
        vp = (b[:, 0] <= b[j, 2]) & (b[j, 0] <= b[:, 2])
        hp = (b[:, 1] <= b[j, 3]) & (b[j, 1] <= b[:, 3])
        both = vp & hp & other
        m = both & top
        nb[m, 0], cur[m, 0] = j, 0
        m = both & ~top & bottom
        nb[m, 2], cur[m, 2] = j, 0
        only_v = vp & ~hp & other
        d_top, d_bot = b[:, 1] - b[j, 3], b[j, 1] - b[:, 3]
        m_top = only_v & top & (height > 2 * cur[:, 0]) & (cur[:, 0] > d_top)       # height / 2 > cur > d
        m_bot = only_v & ~m_top & bottom & (cur[:, 2] > d_bot)
        nb[m_top, 0], cur[m_top, 0] = j, d_top[m_top]
        nb[m_bot, 2], cur[m_bot, 2] = j, d_bot[m_bot]
        only_h = hp & ~vp & other
        d_right, d_left = b[j, 0] - b[:, 2], b[:, 0] - b[j, 2]
        m_r = only_h & right & (width > 2 * cur[:, 1]) & (cur[:, 1] > d_right)
        m_l = only_h & ~m_r & left & (cur[:, 3] > d_left)
        nb[m_r, 1], cur[m_r, 1] = j, d_right[m_r]
        nb[m_l, 3], cur[m_l, 3] = j, d_left[m_l]
    found = nb != ids[:, None]
    sel = np.where(found, nb, -1).astype(np.int64)
    dist = np.where(found, cur, int(max_dist)).astype(np.int64)
    return sel, dist


def _ccw(a, b, c):
    return (c[1] - a[1]) * (b[0] - a[0]) > (b[1] - a[1]) * (c[0] - a[0])


def segments_cross(a, b, c, d):
    """builder.py:358-363 (``intersect``), any common scale of the coordinates"""
    return _ccw(a, c, d) != _ccw(b, c, d) and _ccw(a, b, c) != _ccw(a, b, d)


def _crossed_by_any(v1, v2, h1, h2):
    """does the segment v1-v2 cross any of the segments h1[k]-h2[k] whose END differs from v1 (builder.py:350-379)"""
    if len(h1) == 0:
        return False
    ccw = lambda a, b_, c: (c[..., 1] - a[..., 1]) * (b_[..., 0] - a[..., 0]) > (b_[..., 1] - a[..., 1]) * (c[..., 0] - a[..., 0])
    a, b_ = v1[None, :], v2[None, :]
    cross = (ccw(a, h1, h2) != ccw(b_, h1, h2)) & (ccw(a, b_, h1) != ccw(a, b_, h2))
    differs = (h2[:, 0] != v1[0]) | (h2[:, 1] != v1[1])
    return bool((cross & differs).any())


def crossing_removed(sel, bboxs):
    """sel with the vertical entries (columns 0, 2) that remove_vertical() drops set to -1."""
    b = np.asarray(bboxs, dtype=np.int64).reshape(-1, 4)
    c2 = centres2(b) if len(b) else np.zeros((0, 2), dtype=np.int64)
    n = len(sel)
    h_edges = [(int(sel[j, 3]), j) for j in range(n) if sel[j, 3] >= 0] + [(j, int(sel[j, 1])) for j in range(n) if sel[j, 1] >= 0]
    h1 = c2[[e[0] for e in h_edges]] if h_edges else np.zeros((0, 2), dtype=np.int64)
    h2 = c2[[e[1] for e in h_edges]] if h_edges else np.zeros((0, 2), dtype=np.int64)
    out = sel.copy()
    for i in range(n):
        for p, (s_, d) in ((0, (int(sel[i, 0]), i)), (2, (i, int(sel[i, 2])))):
            if sel[i, p] < 0:
                continue
            if _crossed_by_any(c2[s_], c2[d], h1, h2):
                out[i, p] = -1
    return out


def visibility_edges(bboxs, size, max_dist=500):
    """(u, v) of get_edges('visibility') in the reference's order: surviving vertical edges, then horizontal edges."""
    sel, _ = visibility_select(bboxs, size, max_dist)
    n = len(sel)
    v_edges, h_edges = [], []
    vset, hset = set(), set()
    for i in range(n):
        t, r, bt, l = (int(x) for x in sel[i])
        if t >= 0 and (i, t) not in vset:                    # pos 0: top
            v_edges.append((t, i)); vset.add((t, i))
        if r >= 0 and (r, i) not in hset:                    # pos 1: right
            h_edges.append((i, r)); hset.add((i, r))
        if bt >= 0 and (bt, i) not in vset:                  # pos 2: bottom
            v_edges.append((i, bt)); vset.add((i, bt))
        if l >= 0 and (i, l) not in hset:                    # pos 3: left
            h_edges.append((l, i)); hset.add((l, i))
    b = np.asarray(bboxs, dtype=np.int64)
    c2 = centres2(b) if n else np.zeros((0, 2), dtype=np.int64)
    h1 = c2[[e[0] for e in h_edges]] if h_edges else np.zeros((0, 2), dtype=np.int64)
    h2 = c2[[e[1] for e in h_edges]] if h_edges else np.zeros((0, 2), dtype=np.int64)
    keep = [(s_, d) for (s_, d) in v_edges if not _crossed_by_any(c2[s_], c2[d], h1, h2)]
    edges = keep + h_edges
    u = np.asarray([e[0] for e in edges], dtype=np.int64)
    v = np.asarray([e[1] for e in edges], dtype=np.int64)
    return u, v


def write_reference_fixture(path):
    """Build container only (needs /root/reference): the reference's own (u, v) for seeded pages."""
    from .knn_graph import big_fixture_pages, fixture_pages, reference_get_edges
    out = {}
    pages = fixture_pages(seed=11) + big_fixture_pages(seed=29)
    for i, (b, size, _, max_dist) in enumerate(pages):
        u, v = reference_get_edges(b, size, 5, max_dist, mode="visibility")
        out[f"bbox{i}"], out[f"size{i}"] = b.astype(np.int32), np.asarray(size, dtype=np.int32)
        out[f"maxd{i}"] = np.int32(max_dist)
        out[f"u{i}"], out[f"v{i}"] = u.astype(np.int32), v.astype(np.int32)
    out["n_pages"] = np.int32(len(pages))
    np.savez_compressed(path, **out)
