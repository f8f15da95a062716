"""Export the reference's page graphs to the `.npz` this repository trains from.  Runs on the REFERENCE side (DGL installed, the
reference repository importable as `src`); nothing here is imported by the package, the tests or the GPU box.

    python tools/export_dgl_pages.py --out pages_knn_bbox.npz [the reference's own model_train flags, e.g. --mode=knn --features BBOX]
        from the root of the reference checkout: builds `Papers2Graphs(config)`, runs `modify_graphs` and writes the file.

    python tools/export_dgl_pages.py --bin GRAPHS/train/KNN.bin --info GRAPHS/train/INFO.pkl --out raw.npz
        low-level: the cached graphs as `save_graphs` / `save_info` left them (src/components/graphs/loader.py:98-113), without
        `modify_graphs` -- labels, boxes and (if present) `ndata['feat']` / `edata['feat']` as stored.

What is exported is the contract `train(data, config)` consumes (src/models/model_train.py:213-298), per page i:
  * `data.graphs[i]` AFTER `modify_graphs` (loader.py:206-393): island removal, `to_simple` + `to_bidirected`, edge weights
    `edata['feat'] = 1 - d / max d` (:332-344), converted labels (:346-354);
  * node features = `_generate_features(bboxs, texts, None, create_models(config))` (src/components/graphs/utils.py:9-25; the
    reference recomputes them for every batch of every epoch, model_train.py:293: they are a pure function of the page);
  * the word boxes `data.pages[i]['bboxs']` (post-processing reads them).
File layout = `gnn_tableextraction_amd.components.graphs.loader.PrebuiltPages.save`: concatenated COO (`src`, `dst`, global node
ids), `weight`, `feat`, `label`, `bbox`, per-page `node_off` / `edge_off`, `num_classes`.  Load with `PrebuiltPages.load(path)`.
"""
import argparse
import sys

import numpy as np


def pages_to_npz(graphs, feats, bboxs, num_classes, out):
    """graphs: DGLGraph per page (edges u -> v, ndata['label'], optional edata['feat']); feats[i]: float [n_i, F]; bboxs[i]: [n_i, 4]."""
    src, dst, wgt, feat, lab, box, node_off, edge_off = [], [], [], [], [], [], [0], [0]
    for g, f, b in zip(graphs, feats, bboxs):
        n = int(g.num_nodes())
        u, v = (t.cpu().numpy().astype(np.int64) for t in g.edges())
        f = np.asarray(f, dtype=np.float32)[:n]                       # (as _generate_features trims: page[:lenght_])
        if f.shape[0] != n:
            raise ValueError(f"page {len(node_off) - 1}: {f.shape[0]} feature rows for {n} nodes")
        src.append(u + node_off[-1])
        dst.append(v + node_off[-1])
        w = g.edata['feat'].cpu().numpy().astype(np.float32) if 'feat' in g.edata else np.ones(len(u), dtype=np.float32)
        wgt.append(w)
        feat.append(f)
        lab.append(g.ndata['label'].cpu().numpy().astype(np.int64))
        bb = np.asarray(b, dtype=np.int32).reshape(-1, 4)
        box.append(bb[:n] if len(bb) >= n else np.zeros((n, 4), dtype=np.int32))
        node_off.append(node_off[-1] + n)
        edge_off.append(edge_off[-1] + len(u))
    np.savez(out, src=np.concatenate(src).astype(np.int32), dst=np.concatenate(dst).astype(np.int32), weight=np.concatenate(wgt),
             feat=np.concatenate(feat), label=np.concatenate(lab), bbox=np.concatenate(box), node_off=np.array(node_off, dtype=np.int64),
             edge_off=np.array(edge_off, dtype=np.int64), num_classes=int(num_classes))
    print(f"wrote {out}: {len(graphs)} pages, {node_off[-1]} nodes, {edge_off[-1]} edges, F = {feat[0].shape[1]}")


def export_modified(out, argv):
    """The reference's own pipeline up to the train loop (model_train.py:460-498), then the export."""
    import yaml
    from attrdict import AttrDict
    from src.components.graphs.loader import Papers2Graphs
    from src.components.graphs.utils import _generate_features
    from src.components.features.utils import create_models
    from src.parsers.graphs import parse_args_ModelTrain
    from src.utils.paths import CONFIG
    sys.argv = [sys.argv[0]] + argv                                   # parse_args_ModelTrain reads sys.argv
    with open(CONFIG / 'graph' / "empty.yaml") as f:
        config = AttrDict(parse_args_ModelTrain(AttrDict(yaml.safe_load(f))))
    data = Papers2Graphs(config=config)
    data.modify_graphs(num_graphs=config.TRAINING.num_graphs)
    models = create_models(config)
    bboxs = [p['bboxs'] for p in data.pages]
    texts = [p['texts'] for p in data.pages]
    feats, step = [], 100                                             # (the embedders batch over pages: bounded memory)
    for i in range(0, len(data.graphs), step):
        feats.extend(t.float().cpu().numpy() for t in _generate_features(bboxs[i:i + step], texts[i:i + step], None, models))
    pages_to_npz(data.graphs, feats, bboxs, data.num_classes, out)


def export_cached(bin_path, info_path, out):
    from dgl import load_graphs
    from dgl.data.utils import load_info
    graphs, _ = load_graphs(bin_path)
    info = load_info(info_path)
    feats = [g.ndata['feat'].float().cpu().numpy() if 'feat' in g.ndata else np.zeros((g.num_nodes(), 0), dtype=np.float32) for g in graphs]
    pages_to_npz(graphs, feats, [p['bboxs'] for p in info['pages']], info['num_classes'], out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", required=True)
    ap.add_argument("--bin")
    ap.add_argument("--info")
    args, rest = ap.parse_known_args()
    if args.bin:
        export_cached(args.bin, args.info, args.out)
    else:
        export_modified(args.out, rest)
