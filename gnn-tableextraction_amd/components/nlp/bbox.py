"""BBOX embedder on the device: the interface of the reference's ``Bbox`` (src/components/nlp/bbox.py:17-124:
``_online_batch_(bboxs, texts, titles)`` -> one [n_p, 13] tensor per page, called per batch through ``_generate_features`` at
src/models/model_train.py:293), with the per-word Python loops replaced by ONE launch of gte_bbox_features over all pages.

Host side: only the character classification (``str.isalpha`` / ``str.isdigit`` are Unicode tables).  Everything arithmetic --
width, height, centre with Python's int(w/2) truncation, area, the float64 histogram with "largest bin absorbs 1 - sum" --
runs in csrc/batch_ops.hip and is bit-exact against the reference's own functions (tests/test_aux_golden.py)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from ... import graph as G


def char_class_counts(texts: Sequence[str]) -> np.ndarray:
    """int32 [n, 3]: (#letters, #digits, #others) of every word with spaces removed (bbox.py:79-87: isalpha, else isdigit,
    else other)."""
    out = np.zeros((len(texts), 3), dtype=np.int32)
    for i, t in enumerate(texts):
        a = d = o = 0
        for ch in t.replace(" ", ""):
            if ch.isalpha():
                a += 1
            elif ch.isdigit():
                d += 1
            else:
                o += 1
        out[i] = (a, d, o)
    return out


class Bbox:
    """Drop-in for the reference's BBOX embedder: ``Bbox(device)(bboxs, texts, titles) -> List[Tensor[n_p, 13]]``."""

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)

    def _online_batch_(self, bboxs, texts, titles=None):
        return self.__call__(bboxs, texts, titles)

    def features(self, bboxs, texts) -> torch.Tensor:
        """All pages at once: float32 [sum n_p, 13] on the device."""
        sizes = [len(b) for b in bboxs]
        if sum(sizes) == 0:
            return torch.zeros((0, 13), dtype=torch.float32, device=self.device)
        flat = np.concatenate([np.asarray(b, dtype=np.int32).reshape(-1, 4) for b in bboxs])
        words: List[str] = [w for page in texts for w in page]
        counts = char_class_counts(words)
        return G.bbox_features(torch.from_numpy(flat).to(self.device), torch.from_numpy(counts).to(self.device))

    def __call__(self, bboxs, texts, titles=None, split=None):
        feats = self.features(bboxs, texts)
        return list(torch.split(feats, [len(b) for b in bboxs]))
