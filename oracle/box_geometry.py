"""CPU oracle for the box geometry either side of the model  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

  distance()      restates ``distance(rectA, rectB)`` of src/components/graphs/utils.py:56-88: 0 when the boxes intersect
                  (touching counts: the comparisons are <=), the axis gap when they face each other, int(sqrt(dx^2 + dy^2))
                  between the nearest corners otherwise.  Integer pixel coordinates, integer result.
  edge_weights()  restates the loop of src/components/graphs/loader.py:332-344: d_e = distance(bbox[src_e], bbox[dst_e]),
                  w_e = 1 - d_e / max_e d  (float64 arithmetic), stored as float32.

Pinned by tests/golden/aux_box_distance.npz and aux_edge_weights.npz, which oracle/make_aux_golden.py wrote by running
the reference's own ``distance`` (cut out of its file with ``ast``) on seeded boxes incl. touching / identical /
corner-to-corner cases: tests/test_aux_golden.py.
"""
from math import sqrt

import numpy as np


def distance(a, b) -> int:
    """Scalar form, statement by statement after graphs/utils.py:56-88."""
    left = (b[2] - a[0]) <= 0
    bottom = (a[3] - b[1]) <= 0
    right = (a[2] - b[0]) <= 0
    top = (b[3] - a[1]) <= 0
    if (a[0] <= b[2] and b[0] <= a[2]) and (a[1] <= b[3] and b[1] <= a[3]):
        return 0
    if top and left:
        return int(sqrt((b[2] - a[0]) ** 2 + (b[3] - a[1]) ** 2))
    if left and bottom:
        return int(sqrt((b[2] - a[0]) ** 2 + (b[1] - a[3]) ** 2))
    if bottom and right:
        return int(sqrt((b[0] - a[2]) ** 2 + (b[1] - a[3]) ** 2))
    if right and top:
        return int(sqrt((b[0] - a[2]) ** 2 + (b[3] - a[1]) ** 2))
    if left:
        return int(a[0] - b[2])
    if right:
        return int(b[0] - a[2])
    if bottom:
        return int(b[1] - a[3])
    if top:
        return int(a[1] - b[3])
    raise ValueError("unreachable for well-formed boxes (the reference returns inf here)")


def distance_many(a, bs) -> np.ndarray:
    """distance(a, b) for one box against many (int64 [n]); the vector form of the same case analysis."""
    a = np.asarray(a, dtype=np.int64)
    bs = np.asarray(bs, dtype=np.int64).reshape(-1, 4)
    dx = np.maximum(np.maximum(bs[:, 0] - a[2], a[0] - bs[:, 2]), 0)
    dy = np.maximum(np.maximum(bs[:, 1] - a[3], a[1] - bs[:, 3]), 0)
    d = np.maximum(dx, dy)
    diag = (dx > 0) & (dy > 0)
    d[diag] = np.sqrt((dx[diag] ** 2 + dy[diag] ** 2).astype(np.float64)).astype(np.int64)
    return d


def edge_weights(bbox, src, dst) -> np.ndarray:
    """float32 [E] of loader.py:332-344 for ONE page (the maximum is per page)."""
    d = np.array([distance(bbox[int(u)], bbox[int(v)]) for u, v in zip(src, dst)], dtype=np.int64)
    m = int(d.max())
    if m == 0:
        # the reference divides by zero here (loader.py:341-342 raises ZeroDivisionError); the product writes weight 1
        return np.ones(len(d), dtype=np.float32)
    return np.array([1 - int(x) / m for x in d], dtype=np.float64).astype(np.float32)
