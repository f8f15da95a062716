"""Inference entry point on the HIP path: ``test(data, config)``.

Restates ``src/models/model_predict.py:35-174``: rebuild ``GcnSAGE`` from the config, load the best
weights ``WEIGHTS/{logs}.pt`` written by ``EarlyStopping`` (``src/utils/training.py:49``), run the forward
on every page, collect the predictions and per-class precision / recall / F1.  The reference runs ONE forward per
page (launch-latency bound at ~10^2-10^3 nodes); here pages are batched ``config.TRAINING.batch_size`` at a
time (block-diagonal batching does not change a page's logits -- tested bitwise) and split per page after
the arg-max.

Output contract (what post-processing consumes):
  ``{output}/all_pred/{logs}``        pickle of ONE FLAT python list of ints, the pages' node predictions concatenated in
                                      ``data.graphs`` order -- exactly ``pickle.dump(all_pred, open(OUTPUT / 'all_pred' / logs,
                                      'wb'))`` of model_predict.py:151,172-174; ``postprocessing.py:199-216`` slices it
                                      back into pages by ``graph.num_nodes()``
  ``{output}/predictions/{logs}.pkl`` extra, not in the reference: ``{'all_pred': [per-page lists], 'num_nodes': [...]}``
The printed accuracy is the reference's "Mean Test Accuracy": the MEAN OVER PAGES of the per-page accuracy (:150,163),
not the node-weighted accuracy (returned as ``accuracy_nodes``).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F

from .. import graph as G
from ..components.features.utils import calculate_hidden, get_in_feats_
from ..components.graphs.models import GcnSAGE
from ..utils.config import logs_from_config


def hidden_width(config, in_feats, n_classes):
    mode = config.TRAINING.mode_params
    if mode == 'fixed':
        return config.MODES.fixed.h_layer_dim
    if mode == 'scaled':
        return calculate_hidden(in_feats, n_classes, config.MODES.scaled.params_no, config.TRAINING.n_layers)
    return in_feats / 2


def per_class_prf(y_true: np.ndarray, y_pred: np.ndarray, n_classes: int):
    conf = np.zeros((n_classes, n_classes), dtype=np.float64)
    np.add.at(conf, (y_true, y_pred), 1.0)
    tp = np.diag(conf)
    pred_n, true_n = conf.sum(0), conf.sum(1)
    p = np.where(pred_n > 0, tp / np.maximum(pred_n, 1), 0.0)
    r = np.where(true_n > 0, tp / np.maximum(true_n, 1), 0.0)
    f1 = np.where(p + r > 0, 2 * p * r / np.maximum(p + r, 1e-30), 0.0)
    return p, r, f1, conf


def test(data, config, weights_path=None, save_predictions=True):
    if not (config.TRAINING.gpu >= 0 and torch.cuda.is_available()):
        raise RuntimeError("model_predict runs on the MI355X HIP path only (no CPU fallback)")
    device = torch.device('cuda', config.TRAINING.gpu)
    n_classes = data.num_classes
    in_feats = get_in_feats_(config)
    logs = logs_from_config(config)
    out_root = config.GENERAL.get('output_dir', 'output')
    weights_path = weights_path or os.path.join(out_root, 'weights', f'{logs}.pt')

    model = GcnSAGE(in_feats, int(hidden_width(config, in_feats, n_classes)), n_classes, config.TRAINING.n_layers,
                    F.relu, config.TRAINING.dropout)
    model.load_state_dict(torch.load(weights_path, map_location='cpu'))
    model = model.to(device).eval()

    bs = max(1, int(config.TRAINING.batch_size))
    all_pred, all_true = [], []
    mean_test_acc = 0.0
    with torch.no_grad():
        for b0 in range(0, len(data.graphs), bs):
            pages = [g.to(device) for g in data.graphs[b0:b0 + bs]]
            bg = G.batch(pages)
            pred = model(bg).argmax(dim=1).cpu().numpy()
            off = np.cumsum([0] + [g.num_nodes() for g in pages])
            for i, g in enumerate(pages):
                pp, tt = pred[off[i]:off[i + 1]], g.ndata['label'].long().cpu().numpy()
                all_pred.append(pp)
                all_true.append(tt)
                mean_test_acc += float((pp == tt).sum()) / max(g.num_nodes(), 1)          # per-page accuracy (:150)
    y_pred, y_true = np.concatenate(all_pred), np.concatenate(all_true)
    p, r, f1, conf = per_class_prf(y_true, y_pred, n_classes)
    acc_nodes = float((y_pred == y_true).mean()) if len(y_true) else 0.0
    acc = mean_test_acc / max(len(data.graphs), 1)
    print("Mean Test Accuracy {:.4f}".format(acc))                                        # :163
    print(" -> node accuracy {:.4f} | macro-F1 {:.4f}".format(acc_nodes, float(f1.mean())))
    flat = [int(v) for a in all_pred for v in a.tolist()]
    if save_predictions:
        ap_dir = os.path.join(out_root, 'all_pred')                                       # OUTPUT / 'all_pred' (:172)
        os.makedirs(ap_dir, exist_ok=True)
        with open(os.path.join(ap_dir, logs), 'wb') as f:
            pickle.dump(flat, f)                                                          # the reference's artefact (:174)
        pred_dir = os.path.join(out_root, 'predictions')
        os.makedirs(pred_dir, exist_ok=True)
        with open(os.path.join(pred_dir, f'{logs}.pkl'), 'wb') as f:
            pickle.dump({'all_pred': [a.tolist() for a in all_pred], 'num_nodes': [len(a) for a in all_pred]}, f)
    return {'accuracy': acc, 'accuracy_nodes': acc_nodes, 'precision': p, 'recall': r, 'f1': f1, 'confusion': conf,
            'all_pred': all_pred, 'all_pred_flat': flat}
