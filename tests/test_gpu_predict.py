"""Inference entry point: ``test(data, config)`` against the CPU oracle and sklearn."""
import os
import pickle

import numpy as np
import pytest
import torch

from gnn_tableextraction_amd import GcnSAGE
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.components.graphs.loader import PrebuiltPages
from gnn_tableextraction_amd.models import model_predict
from gnn_tableextraction_amd.parsers.graphs import parse_args_ModelTrain
from gnn_tableextraction_amd.utils.config import logs_from_config
from oracle import gcnsage_cpu as oc

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, bs):
    return parse_args_ModelTrain(argv=["--mode=knn", "--features", "BBOX", "--n_layers=3", "--mode_params=fixed", "--h_layer_dim=64",
                                      f"--batch_size={bs}", "--n_epochs=1", "--output_dir", str(tmp_path)])


def _oracle_logits(state, page):
    g = oc.OracleGraph(page.src, page.dst, page.num_nodes, page.weight)
    return oc.gcnsage_forward(state, g, torch.from_numpy(page.feat)).numpy()


@pytest.mark.parametrize("bs", [1, 8])
def test_predictions_equal_the_oracle_argmax_and_the_metrics_equal_sklearn(tmp_path, bs):
    """``all_pred`` of test() == argmax of oracle.gcnsage_forward on the same weights and pages (nodes whose two largest oracle
    logits are closer than 1e-4 are excluded and must be rare); accuracy / precision / recall / F1 against sklearn on those
    predictions; the flat pickle at {output}/all_pred/{logs} is what post-processing reads (model_predict.py:151,172-174)."""
    from sklearn.metrics import precision_recall_fscore_support
    torch.manual_seed(3)
    data = PrebuiltPages.synthetic(24, in_feats=13)
    # BBOX features are raw pixels / areas up to ~1e5: un-normalised, as the reference feeds them
    cfg = _cfg(tmp_path, bs)
    model = GcnSAGE(13, 64, 9, 3, torch.nn.functional.relu, 0)
    logs = logs_from_config(cfg)
    os.makedirs(tmp_path / "weights", exist_ok=True)
    torch.save(model.state_dict(), tmp_path / "weights" / f"{logs}.pt")
    out = model_predict.test(data, cfg)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    n_tie = n_all = 0
    y_true, y_pred = [], []
    for i, page in enumerate(data.page_arrays):
        lo = _oracle_logits(state, page)
        top2 = np.sort(lo, axis=1)[:, -2:]
        sure = (top2[:, 1] - top2[:, 0]) > 1e-4 * np.maximum(1.0, np.abs(top2[:, 1]))
        np.testing.assert_array_equal(out["all_pred"][i][sure], lo.argmax(1)[sure])
        n_tie += int((~sure).sum()); n_all += len(sure)
        y_true.append(page.label); y_pred.append(out["all_pred"][i])
    assert n_tie <= 0.01 * n_all
    y_true, y_pred = np.concatenate(y_true), np.concatenate(y_pred)
    p, r, f1, _ = precision_recall_fscore_support(y_true, y_pred, labels=list(range(9)), zero_division=0)
    np.testing.assert_allclose(out["precision"], p, atol=1e-12)
    np.testing.assert_allclose(out["recall"], r, atol=1e-12)
    np.testing.assert_allclose(out["f1"], f1, atol=1e-12)
    per_page = [float((out["all_pred"][i] == pg.label).mean()) for i, pg in enumerate(data.page_arrays)]
    assert abs(out["accuracy"] - float(np.mean(per_page))) < 1e-12
    assert abs(out["accuracy_nodes"] - float((y_true == y_pred).mean())) < 1e-12
    flat = pickle.load(open(tmp_path / "all_pred" / logs, "rb"))
    assert flat == y_pred.tolist()


def test_page_forward_graphs_equal_the_eager_forward_bitwise():
    """One HIP-graph launch per page (size buckets, the page assembled straight into the bucket's buffers) gives the logits of
    the eager forward of that page, bit for bit, whatever page used the bucket before."""
    torch.manual_seed(5)
    data = PrebuiltPages.synthetic(20, in_feats=13)
    dev = torch.device("cuda:0")
    model = GcnSAGE(13, 64, 9, 3, torch.nn.functional.relu, 0).to(dev).eval()
    res = G.ResidentPages(data.graphs, dev)
    runner = model_predict.PageForwardGraphs(model, res)
    order = [3, 0, 7, 7, 19, 1, 12, 3]
    for pid in order:
        logits, pred = runner.forward(pid)
        with torch.no_grad():
            want = model(data.graphs[pid].to(dev))
        assert torch.equal(logits, want) and torch.equal(pred, want.argmax(1))
    assert len(runner._b) >= 1
