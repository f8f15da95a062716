"""CPU oracle for the BBOX node features  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates ``Bbox.__call__`` of the reference, src/components/nlp/bbox.py:49-124 (``get_shape`` :49-54,
``get_histogram`` :56-107), in plain Python with the same operations in the same order (float64 histogram,
"largest bin absorbs 1 - sum", empty text -> [0, 0, 0, 1]).  The reference holds no test or golden vector for
it; the restatement is pinned by tests/golden/aux_bbox_features.npz -- 1 500 words (ASCII and non-ASCII letters / digits,
empty and blank texts, degenerate boxes) run through the reference's own nested ``get_shape`` / ``get_histogram``
(ast-extracted by oracle/make_aux_golden.py) -- bit for bit (tests/test_aux_golden.py), plus hand cases in
tests/test_bbox_features.py.
"""
import numpy as np


def char_counts(text: str):
    """(#letters, #digits, #others) of the characters of ``text`` without spaces (bbox.py:79-87)."""
    a = d = o = 0
    for ch in text.replace(" ", ""):
        if ch.isalpha():
            a += 1
        elif ch.isdigit():
            d += 1
        else:
            o += 1
    return a, d, o


def histogram(counts):
    na, nd, no = (int(c) for c in counts)
    ns = na + nd + no
    h = [0.0, 0.0, 0.0, 0.0]
    if ns != 0:
        h[0], h[1], h[2] = na / ns, nd / ns, no / ns
        if sum(h) != 1.0:
            diff = 1.0 - sum(h)
            m = max(h) + diff
            h[h.index(max(h))] = m
    if h[:3] == [0.0, 0.0, 0.0]:
        h[3] = 1.0
    return h


def shape(b):
    w, hgt = b[2] - b[0], b[3] - b[1]
    return [w, hgt, b[2] - int(w / 2), b[3] - int(hgt / 2), w * hgt, b[0], b[1], b[2], b[3]]


def bbox_features(bboxes, counts) -> np.ndarray:
    """float32 [N, 13] exactly as ``np.append(emb_shape, emb_hist, 1)`` -> torch.tensor -> .float()."""
    rows = [shape([int(v) for v in b]) + histogram(c) for b, c in zip(bboxes, counts)]
    return np.asarray(rows, dtype=np.float64).astype(np.float32).reshape(-1, 13)
