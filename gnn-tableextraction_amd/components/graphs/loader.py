"""Pre-built page-graph dataset: the OUTPUT CONTRACT of the reference's ``Papers2Graphs`` after
``modify_graphs`` (src/components/graphs/loader.py:206-393) without the PDF pipeline behind it.

The train loop needs from ``data``: ``num_classes``, ``stats``, ``graphs`` (each with
``ndata['feat'|'label']`` and ``edata['feat']``), ``split(n)`` (95/5, ``random.sample`` seeded 42:
loader.py:48,395-420), ``label_tranformer`` and ``__len__``.  DGL's ``.bin`` cache cannot be read
without DGL, so the on-disk format here is one ``.npz`` holding the concatenated COO, features,
labels and per-page offsets (``save`` / ``load``).
"""
from __future__ import annotations

import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from ...graph import PageGraph
from ...data import synthetic as S

# src/components/graphs/labels.py:13-19: 13 categories minus the never-instantiated {4, 9, 11, 12}
ORIGIN_TO_CONV = {0: 0, 1: 1, 2: 2, 3: 3, 5: 4, 6: 5, 7: 6, 8: 7, 10: 8, 4: None, 9: None, 11: None, 12: None}


class LabelTransformer:
    def __init__(self):
        self.origin_to_conv = dict(ORIGIN_TO_CONV)
        self.conv_to_origin = {v: k for k, v in ORIGIN_TO_CONV.items() if v is not None}


class PrebuiltPages:
    def __init__(self, pages: Sequence[S.Page], num_classes: int = 9, rate: float = 0.95, seed: int = 42,
                 float_labels: bool = True):
        self.page_arrays = list(pages)
        self.num_classes = num_classes
        self.rate, self.seed = rate, seed
        self.label_tranformer = LabelTransformer()          # (sic) the reference's attribute name
        self.graphs: List[PageGraph] = []
        counts = np.zeros(num_classes, dtype=np.int64)
        for p in self.page_arrays:
            g = PageGraph(p.src, p.dst, p.num_nodes)
            g.ndata['feat'] = torch.from_numpy(p.feat)
            # the reference stores labels as float32 and casts at the loss (loader.py:350-354)
            g.ndata['label'] = torch.from_numpy(p.label.astype(np.float32) if float_labels else p.label)
            g.edata['feat'] = torch.from_numpy(p.weight)
            self.graphs.append(g)
            counts += np.bincount(p.label, minlength=num_classes)
        total = max(int(counts.sum()), 1)
        self.stats = {'numbers': counts.tolist(), 'percentages': [round(c / total, 2) for c in counts.tolist()]}
        self.pages = [{'page': f'synthetic_{i}', 'bboxs': p.bbox, 'texts': None} for i, p in enumerate(self.page_arrays)]

    def __len__(self):
        return len(self.graphs)

    def split(self, num_graphs: Optional[int] = None):
        n = min(num_graphs or len(self), len(self))
        train_amount = int(n * self.rate)
        rnd = random.Random(self.seed)
        train_idx = rnd.sample(range(0, n), train_amount)
        val_idx = sorted(set(range(0, n)) - set(train_idx))
        return [self.graphs[i] for i in train_idx], [self.graphs[i] for i in val_idx], (train_idx, val_idx)

    # ---- on-disk format replacing save_graphs(.bin) + save_info(.pkl) (loader.py:98-113) ----------
    def save(self, path: str) -> None:
        src, dst, w, feat, label, off = S.concat_pages(self.page_arrays)
        eoff = np.cumsum([0] + [len(p.src) for p in self.page_arrays])
        np.savez(path, src=src, dst=dst, weight=w, feat=feat, label=label, node_off=off, edge_off=eoff,
                 bbox=np.concatenate([p.bbox for p in self.page_arrays]), num_classes=self.num_classes)

    @classmethod
    def load(cls, path: str, **kw) -> "PrebuiltPages":
        z = np.load(path)
        no, eo = z["node_off"], z["edge_off"]
        pages = []
        for i in range(len(no) - 1):
            n0, n1, e0, e1 = int(no[i]), int(no[i + 1]), int(eo[i]), int(eo[i + 1])
            pages.append(S.Page((z["src"][e0:e1] - n0).astype(np.int32), (z["dst"][e0:e1] - n0).astype(np.int32),
                                z["weight"][e0:e1], z["feat"][n0:n1], z["label"][n0:n1], z["bbox"][n0:n1]))
        return cls(pages, num_classes=int(z["num_classes"]), **kw)

    @classmethod
    def from_boxes(cls, page_boxes, page_sizes, page_texts, page_labels, device, k: int = 5, max_dist: int = 500,
                   bidirectional: bool = True, range_island: int = 0, extra_feats=None, mode: str = "knn",
                   **kw) -> "PrebuiltPages":
        """Pages from word boxes with the graph stage on the DEVICE: k-NN edges, island removal,
        to_simple + to_bidirected, edge weights (graph.knn_graph_from_boxes) and the 13 BBOX node features
        (graph.bbox_features) -- what builder.get_graph + loader.modify_graphs + nlp/bbox.py do per page in Python.
        ``page_boxes[p]`` int [n_p, 4], ``page_sizes[p]`` = (width, height), ``page_texts[p]`` the words (for the character
        histogram; None -> empty texts), ``page_labels[p]`` converted class ids, ``extra_feats[p]`` optional float32 [n_p, F']
        columns appended after the BBOX features (REPR / SPACY / SCIBERT embeddings stay pre-computed)."""
        from ... import graph as G
        from ..nlp.bbox import Bbox
        device = torch.device(device)
        n_pages = len(page_boxes)
        sizes = [len(b) for b in page_boxes]
        node_off = np.concatenate([[0], np.cumsum(sizes)])
        bbox = torch.from_numpy(np.concatenate([np.asarray(b, dtype=np.int32).reshape(-1, 4) for b in page_boxes])).to(device)
        labels = torch.from_numpy(np.concatenate([np.asarray(l, dtype=np.int64) for l in page_labels])).to(device)
        g, keep = G.knn_graph_from_boxes(bbox, node_off, np.asarray(page_sizes, dtype=np.int32), k=k, max_dist=max_dist,
                                         bidirectional=bidirectional, labels=labels, range_island=range_island,
                                         mode=mode)                  # PREPROCESS.mode: 'knn' | 'visibility' (loader.py:80)
        texts = page_texts if page_texts is not None else [[""] * n for n in sizes]
        feat = Bbox(device).features(page_boxes, texts)[keep]
        if extra_feats is not None:
            extra = torch.from_numpy(np.concatenate([np.asarray(e, dtype=np.float32) for e in extra_feats])).to(device)[keep]
            feat = torch.cat([feat, extra], dim=1)
        # per-page host arrays (the dataset object's contract: .graphs[i] with ndata feat/label, edata feat, .pages[i]['bboxs'])
        src, dst = (t.cpu().numpy() for t in g.edges())
        w = g.edata['feat'].cpu().numpy()
        feat_h, lab_h, bbox_h = feat.cpu().numpy(), g.ndata['label'].cpu().numpy(), g.ndata['bbox'].cpu().numpy()
        off = np.concatenate([[0], np.cumsum(g.batch_num_nodes_)])
        eoff = np.searchsorted(dst, off)                   # edges are sorted by destination
        pages = []
        for p in range(n_pages):
            n0, n1, e0, e1 = int(off[p]), int(off[p + 1]), int(eoff[p]), int(eoff[p + 1])
            pages.append(S.Page((src[e0:e1] - n0).astype(np.int32), (dst[e0:e1] - n0).astype(np.int32), w[e0:e1],
                                np.ascontiguousarray(feat_h[n0:n1]), lab_h[n0:n1].astype(np.int64), bbox_h[n0:n1]))
        out = cls(pages, **kw)
        g.ndata['feat'] = feat
        out.whole = g                                       # the batched device graph, for callers that stay on the device
        return out

    @classmethod
    def synthetic(cls, n_pages: int, in_feats: int = 13, **kw) -> "PrebuiltPages":
        return cls(S.make_pages(n_pages, in_feats=in_feats), **kw)
