"""Golden vectors for the steps either side of the model  --  TEST INFRASTRUCTURE, run ONLY in the build container.

Runs the REFERENCE's own code (extracted with ``ast`` from /root/reference, never copied into this repository, never
shipped to the GPU box) on seeded inputs and writes small fixtures under tests/golden/:

  aux_box_distance.npz   ``distance(rectA, rectB)``              src/components/graphs/utils.py:56-88
  aux_edge_weights.npz   the edge-weight loop ``1 - d / max(d)``  src/components/graphs/loader.py:332-344
  aux_bbox_features.npz  nested ``get_shape`` / ``get_histogram`` + the np.append / torch.tensor / .float() around them
                                                                  src/components/nlp/bbox.py:49-124, model_train.py:296
  aux_kats.json          ``EarlyStopping.step`` on scripted loss series   src/utils/training.py:14-49
                         ``LableModification`` maps                      src/components/graphs/labels.py:7-27
                         ``calculate_hidden`` / ``get_in_feats_`` answers  src/components/features/utils.py:71-101
  aux_knn_edges.npz      the k-NN edge builder of ``GraphBuilder.get_graph`` (mode 'knn')  builder.py:240-292
  aux_visibility_edges.npz   the same function in mode 'visibility' (nearest visible box per direction, crossing
                         vertical edges removed)  builder.py:294-379
                         on seeded pages (ties included; how they are compared: oracle/knn_graph.py)

The modules these live in import packages that are absent here (dgl, seaborn, attrdict, fitz, a hard-coded path check
in src/utils/paths.py), so the functions are cut out of their files by name and compiled alone, with exactly the
globals their bodies use (math.sqrt / inf, numpy, torch).  What is executed is the reference's text, unmodified.

    python oracle/make_aux_golden.py            # writes tests/golden/aux_*
"""
import ast
import json
import os
import sys
from math import inf, sqrt

import numpy as np
import torch

REF = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def extract(path, name, kind=(ast.FunctionDef, ast.ClassDef), glb=None):
    """Compile the first def/class called ``name`` found anywhere in ``path`` (nested defs included) on its own."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, kind) and node.name == name:
            text = ast.get_source_segment(src, node)
            lines = text.split("\n")
            indent = node.col_offset
            text = "\n".join([lines[0]] + [l[indent:] if l[:indent].strip() == "" else l for l in lines[1:]])
            ns = dict(glb or {})
            exec(compile(text, f"{path}:{name}", "exec"), ns)
            return ns[name]
    raise KeyError(f"{name} not found in {path}")


def boxes(rng, n):
    """word-like integer boxes on an A4-at-72dpi-ish canvas, incl. degenerate (zero-size) and duplicated ones"""
    x0, y0 = rng.integers(0, 1600, n), rng.integers(0, 2300, n)
    b = np.stack([x0, y0, x0 + rng.integers(0, 140, n), y0 + rng.integers(0, 40, n)], 1)
    return b.astype(np.int64)


def main(out=OUT):
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(2024)

    # ---- distance() ------------------------------------------------------------------------------------------
    distance = extract("components/graphs/utils.py", "distance", glb={"sqrt": sqrt, "inf": inf})
    a, b = boxes(rng, 6000), boxes(rng, 6000)
    # touching / overlapping / aligned cases on purpose
    b[:500, 0] = a[:500, 2]; b[:500, 2] = b[:500, 0] + 30           # B starts exactly where A ends (x)
    b[500:1000, 1] = a[500:1000, 3]; b[500:1000, 3] = b[500:1000, 1] + 12
    b[1000:1500] = a[1000:1500]                                       # identical boxes
    b[1500:2000, :2] = a[1500:2000, 2:]; b[1500:2000, 2:] = b[1500:2000, :2] + 9   # corner touching
    d = np.array([distance(ra.tolist(), rb.tolist()) for ra, rb in zip(a, b)], dtype=np.int64)
    np.savez_compressed(os.path.join(out, "aux_box_distance.npz"), a=a.astype(np.int32), b=b.astype(np.int32), dist=d)

    # ---- edge weights (loader.py:332-344, the statements inside the `if ... edge_features:` block) ---------------
    pages = []
    for p in range(6):
        n = int(rng.integers(8, 120))
        bb = boxes(rng, n)
        e = int(rng.integers(n, 6 * n))
        u, v = rng.integers(0, n, e), rng.integers(0, n, e)
        srcs, dsts = u.tolist(), v.tolist()
        bboxs = bb.tolist()
        distances = []
        for i, src in enumerate(srcs):
            distances.append(distance(bboxs[src], bboxs[dsts[i]]))
        m = max(distances)
        distances = [(1 - dd / m) for dd in distances]
        w = torch.tensor(distances, dtype=torch.float32).numpy()
        pages.append((bb, u, v, w))
    np.savez_compressed(os.path.join(out, "aux_edge_weights.npz"),
                        n_pages=len(pages),
                        **{f"bbox{i}": p[0].astype(np.int32) for i, p in enumerate(pages)},
                        **{f"src{i}": p[1].astype(np.int32) for i, p in enumerate(pages)},
                        **{f"dst{i}": p[2].astype(np.int32) for i, p in enumerate(pages)},
                        **{f"w{i}": p[3] for i, p in enumerate(pages)})

    # ---- BBOX node features ---------------------------------------------------------------------------------------
    get_shape = extract("components/nlp/bbox.py", "get_shape")
    get_histogram = extract("components/nlp/bbox.py", "get_histogram")
    alphabet = list("abcdefgXYZ0123456789.,;:-()%$ éß中٣")      # letters, digits (incl. non-ASCII), others, space
    texts = []
    for i in range(1500):
        k = int(rng.integers(0, 14))
        texts.append("".join(rng.choice(alphabet, size=k)))
    texts[:6] = ["", " ", "   ", "abc", "123", "a1."]
    page_bbox = boxes(rng, len(texts))
    page_bbox[:40, 2] = page_bbox[:40, 0] - rng.integers(0, 9, 40)     # negative / zero widths: int(w/2) truncates toward 0
    emb_shape = list(map(get_shape, page_bbox.tolist()))
    emb_hist = list(map(get_histogram, texts))
    feats = torch.tensor(np.append(emb_shape, emb_hist, 1)).float().numpy()    # bbox.py:121 + model_train.py:296 (.float())
    counts = np.array([[sum(c.isalpha() for c in t.replace(" ", "")), sum((not c.isalpha()) and c.isdigit() for c in t.replace(" ", "")),
                        0] for t in texts], dtype=np.int64)
    counts[:, 2] = np.array([len(t.replace(" ", "")) for t in texts]) - counts[:, 0] - counts[:, 1]
    np.savez_compressed(os.path.join(out, "aux_bbox_features.npz"), bbox=page_bbox.astype(np.int32),
                        texts=np.array(texts, dtype=np.str_), char_counts=counts.astype(np.int32), feat=feats)

    # ---- scripted-series KATs -------------------------------------------------------------------------------------
    saves = []

    class _Model:
        def state_dict(self):
            return {}
    import datetime
    fake_torch = type("T", (), {"save": staticmethod(lambda sd, path: saves.append(path))})
    EarlyStopping = extract("utils/training.py", "EarlyStopping", glb={"datetime": datetime, "torch": fake_torch, "inf": inf})
    series = {
        "improve_then_plateau": [1.0, 0.9, 0.8, 0.8, 0.85, 0.81, 0.79, 0.9, 0.9, 0.9, 0.9],
        "ties_count_as_improvement": [0.5, 0.5, 0.5, 0.6, 0.5, 0.7, 0.7, 0.7],
        "never_improves": [0.3, 0.4, 0.5, 0.6, 0.7, 0.8],
        "nan_in_the_middle": [1.0, float("nan"), 0.9, 1.1, 1.2, 1.3],
    }
    es = {}
    for name, losses in series.items():
        for patience in (3, 50):
            saves.clear()
            st = EarlyStopping("W", "run", patience=patience)
            trace = []
            for l in losses:
                n0 = len(saves)
                stop, counter = st.step(l, _Model())
                trace.append([bool(stop), int(counter), len(saves) - n0])
            es[f"{name}/p{patience}"] = {"losses": [None if l != l else l for l in losses], "patience": patience,
                                         "trace": trace, "save_path": saves[0] if saves else None}

    class _Cat:       # len(Categories_names): the Enum of src/utils/const.py:4-18 has 13 members
        pass
    const_src = open(os.path.join(REF, "utils/const.py")).read()
    enum_node = next(n for n in ast.walk(ast.parse(const_src)) if isinstance(n, ast.ClassDef) and n.name == "Categories_names")
    names = [t.targets[0].id for t in enum_node.body if isinstance(t, ast.Assign)]
    values = [t.value.value for t in enum_node.body if isinstance(t, ast.Assign)]
    LableModification = extract("components/graphs/labels.py", "LableModification",
                                glb={"np": np, "MAX_CLASS_COUNT": len(names)})
    import yaml
    to_remove = yaml.safe_load(open("/root/reference/configs/graph/empty.yaml"))["LABELS"]["to_remove"]
    lm = LableModification({"LABELS": {"to_remove": to_remove}})
    labels = {"categories": dict(zip(names, values)), "to_remove": list(to_remove),
              "origin_to_conv": {str(k): v for k, v in lm.origin_to_conv.items()},
              "convert_0_12": lm.convert(list(range(13))), "revert_0_8": lm.revert(list(range(9)))}

    calculate_hidden = extract("components/features/utils.py", "calculate_hidden", glb={"np": np, "sqrt": sqrt, "math": __import__("math")})
    ch = [[f0, c, p, l, float(calculate_hidden(f0, c, p, l))]
          for (f0, c, p, l) in [(13, 9, 100000, 3), (831, 9, 100000, 3), (10000, 8, 100000, 3), (63, 9, 100000, 4),
                                (313, 9, 100000, 3), (363, 9, 100000, 2), (781, 9, 50000, 3)]]
    with open(os.path.join(out, "aux_kats.json"), "w") as f:
        json.dump({"early_stopping": es, "labels": labels, "calculate_hidden": ch}, f, indent=1)

    # ---- k-NN edges (builder.py:240-292) ------------------------------------------------------------------------------
    sys.path.insert(0, ROOT)
    from oracle import knn_graph
    knn_graph.write_reference_fixture(os.path.join(out, "aux_knn_edges.npz"))
    # ---- visibility edges (builder.py:294-379) ------------------------------------------------------------------------
    from oracle import visibility_graph
    visibility_graph.write_reference_fixture(os.path.join(out, "aux_visibility_edges.npz"))
    print("wrote", sorted(f for f in os.listdir(out) if f.startswith("aux_")))


if __name__ == "__main__":
    main(*sys.argv[1:])
