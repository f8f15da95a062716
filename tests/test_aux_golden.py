"""The steps either side of the model against vectors produced by the REFERENCE's own functions (ast-extracted in the
build container by oracle/make_aux_golden.py; the fixtures are data, the generating script is committed):

  CPU:  oracle/box_geometry.py, oracle/bbox_features.py vs the fixtures (pins the oracles);
        SURVEY 8(c)(5),(6) KATs: EarlyStopping traces, ReduceLROnPlateau mirroring, label map, calculate_hidden, per-class
        F1 vs sklearn -- the product's host logic against the reference's answers
  GPU:  gte_edge_weights_bbox and gte_bbox_features bit-exact against the fixtures (through the C ABI)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import bbox_features as ob
from oracle import box_geometry as bg
from tests.conftest import GOLDEN_DIR


def _npz(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


KATS = json.load(open(os.path.join(GOLDEN_DIR, "aux_kats.json")))


# ------------------------------------------------------------------------------------------------ oracles, pinned
def test_box_distance_oracle_matches_the_reference_function():
    z = _npz("aux_box_distance.npz")
    a, b, want = z["a"].astype(np.int64), z["b"].astype(np.int64), z["dist"]
    got = np.array([bg.distance(x, y) for x, y in zip(a, b)])
    np.testing.assert_array_equal(got, want)
    for i in range(0, len(a), 97):                                   # the vector form used by the k-NN oracle
        np.testing.assert_array_equal(bg.distance_many(a[i], b[i:i + 50]), [bg.distance(a[i], y) for y in b[i:i + 50]])
    assert (want == 0).sum() > 500 and (want > 0).sum() > 3000      # both regimes are in the fixture


def test_edge_weight_oracle_matches_the_reference_loop():
    z = _npz("aux_edge_weights.npz")
    for i in range(int(z["n_pages"])):
        got = bg.edge_weights(z[f"bbox{i}"].astype(np.int64), z[f"src{i}"], z[f"dst{i}"])
        np.testing.assert_array_equal(got, z[f"w{i}"])             # bit-exact float32


def test_bbox_feature_oracle_matches_the_reference_functions():
    z = _npz("aux_bbox_features.npz")
    texts = [str(t) for t in z["texts"]]
    counts = np.array([ob.char_counts(t) for t in texts])
    np.testing.assert_array_equal(counts, z["char_counts"])
    np.testing.assert_array_equal(ob.bbox_features(z["bbox"], counts), z["feat"])     # bit-exact float32
    assert (np.float64(z["feat"][:, 9:]).sum(1) - 1).__abs__().max() < 1e-6


# ------------------------------------------------------------------------------------------------ host-logic KATs
class _CountingModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(2))


@pytest.mark.parametrize("name", sorted(KATS["early_stopping"]))
def test_early_stopping_trace_equals_the_reference(name, tmp_path, monkeypatch):
    from gnn_tableextraction_amd.utils import training as T
    kat = KATS["early_stopping"][name]
    saves = []
    monkeypatch.setattr(T.torch, "save", lambda sd, path: saves.append(path))
    st = T.EarlyStopping(str(tmp_path), "run", patience=kat["patience"])
    for loss, (stop, counter, n_saves) in zip(kat["losses"], kat["trace"]):
        n0 = len(saves)
        got = st.step(float("nan") if loss is None else loss, _CountingModel())
        assert (bool(got[0]), int(got[1]), len(saves) - n0) == (stop, counter, n_saves)
    assert all(p.endswith("/run.pt") for p in saves)


def test_label_map_equals_the_reference():
    from gnn_tableextraction_amd.components.graphs.loader import ORIGIN_TO_CONV, LabelTransformer
    want = {int(k): v for k, v in KATS["labels"]["origin_to_conv"].items()}
    assert ORIGIN_TO_CONV == want
    lt = LabelTransformer()
    assert [lt.origin_to_conv.get(i) for i in range(13)] == KATS["labels"]["convert_0_12"]
    assert [lt.conv_to_origin.get(i) for i in range(9)] == KATS["labels"]["revert_0_8"]
    from gnn_tableextraction_amd.models.model_train import TABLE_COLH, TABLE_TCELL
    assert (TABLE_TCELL, TABLE_COLH) == (KATS["labels"]["categories"]["TABLE_TCELL"], KATS["labels"]["categories"]["TABLE_COLH"])


def test_calculate_hidden_equals_the_reference():
    from gnn_tableextraction_amd.components.features.utils import calculate_hidden
    for f0, c, p, l, want in KATS["calculate_hidden"]:
        assert calculate_hidden(f0, c, p, l) == want
    assert int(calculate_hidden(13, 9, 100000, 3)) == 218 and int(calculate_hidden(831, 9, 100000, 3)) == 96


def test_lr_schedule_mirror_follows_reduce_on_plateau():
    """train() drives ReduceLROnPlateau('min', factor=0.5) (model_train.py:175,369) through a stand-in optimiser whose lr
    is copied into the engine: on a scripted val-loss series the mirrored lr must equal what the scheduler does to a real
    Adam, epoch by epoch."""
    losses = [1.0, 0.9] + [0.95] * 11 + [0.5] + [0.7] * 12 + [0.69] * 3
    real = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=0.01)
    s_real = torch.optim.lr_scheduler.ReduceLROnPlateau(real, 'min', factor=0.5)
    holder = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.01)
    s_mirror = torch.optim.lr_scheduler.ReduceLROnPlateau(holder, 'min', factor=0.5)
    engine_lr, seen = 0.01, []
    for l in losses:
        s_real.step(l)
        s_mirror.step(l)
        engine_lr = holder.param_groups[0]['lr']                  # what model_train.train does after scheduler.step
        assert engine_lr == real.param_groups[0]['lr']
        seen.append(engine_lr)
    assert seen[0] == 0.01 and seen[-1] == 0.0025 and sorted(set(seen), reverse=True) == [0.01, 0.005, 0.0025]


def test_per_class_f1_equals_sklearn():
    from sklearn.metrics import precision_recall_fscore_support
    from gnn_tableextraction_amd.models.model_predict import per_class_prf
    rng = np.random.default_rng(0)
    for trial in range(5):
        n_classes = 9
        y_true = rng.choice(n_classes, size=4000, p=[0.6, 0.06, 0.05, 0.04, 0.03, 0.12, 0.05, 0.04, 0.01])
        y_pred = np.where(rng.random(4000) < 0.7, y_true, rng.integers(0, n_classes, 4000))
        if trial == 1:
            y_pred[y_pred == 8] = 0                                 # a class that is never predicted (zero_division)
        if trial == 2:
            y_true[y_true == 4] = 1                                 # a class that never occurs
        p, r, f1, conf = per_class_prf(y_true, y_pred, n_classes)
        wp, wr, wf, _ = precision_recall_fscore_support(y_true, y_pred, labels=list(range(n_classes)), zero_division=0)
        np.testing.assert_allclose(p, wp, atol=1e-12)
        np.testing.assert_allclose(r, wr, atol=1e-12)
        np.testing.assert_allclose(f1, wf, atol=1e-12)
        # the train loop's F1 from the all-reduced confusion matrix (model_train.py): 2 tp / (pred + true)
        tp, denom = np.diag(conf), conf.sum(0) + conf.sum(1)
        np.testing.assert_allclose(np.where(denom > 0, 2 * tp / np.maximum(denom, 1), 0.0), wf, atol=1e-12)


# ------------------------------------------------------------------------------------------------ HIP, through the C ABI
@pytest.mark.gpu
def test_edge_weight_kernel_is_bit_exact_on_the_reference_vectors():
    from gnn_tableextraction_amd import graph as G
    z = _npz("aux_edge_weights.npz")
    n_pages = int(z["n_pages"])
    bbox = np.concatenate([z[f"bbox{i}"] for i in range(n_pages)]).astype(np.int32)
    off = np.cumsum([0] + [len(z[f"bbox{i}"]) for i in range(n_pages)])
    src = np.concatenate([z[f"src{i}"] + off[i] for i in range(n_pages)]).astype(np.int32)
    dst = np.concatenate([z[f"dst{i}"] + off[i] for i in range(n_pages)]).astype(np.int32)
    gon = np.repeat(np.arange(n_pages), np.diff(off)).astype(np.int32)
    want = np.concatenate([z[f"w{i}"] for i in range(n_pages)])
    d = lambda a: torch.from_numpy(a).cuda()
    got = G.edge_weights_from_boxes(d(bbox), d(src), d(dst), d(gon), n_pages).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_box_distance_kernel_is_bit_exact_on_the_reference_vectors():
    """d = (1 - w) * max d recovers the integer distances the kernel computed: one 'page' holding every fixture pair."""
    from gnn_tableextraction_amd import graph as G
    z = _npz("aux_box_distance.npz")
    a, b, want = z["a"], z["b"], z["dist"]
    n = len(a)
    bbox = np.concatenate([a, b]).astype(np.int32)
    src, dst = np.arange(n, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32)
    d = lambda x: torch.from_numpy(x).cuda()
    w = G.edge_weights_from_boxes(d(bbox), d(src), d(dst), d(np.zeros(2 * n, np.int32)), 1).cpu().numpy()
    m = int(want.max())
    np.testing.assert_array_equal(w, (1.0 - want.astype(np.float64) / m).astype(np.float32))


@pytest.mark.gpu
def test_bbox_feature_kernel_is_bit_exact_on_the_reference_vectors():
    from gnn_tableextraction_amd import _lib
    z = _npz("aux_bbox_features.npz")
    n = len(z["bbox"])
    lib = _lib.load()
    b, c = torch.from_numpy(z["bbox"]).cuda(), torch.from_numpy(z["char_counts"]).cuda()
    out = torch.zeros(n, 13, device="cuda")
    _lib.check(lib.gte_bbox_features(_lib.ptr(b), _lib.ptr(c), _lib.ptr(out), 13, n, _lib.current_stream()))
    np.testing.assert_array_equal(out.cpu().numpy(), z["feat"])
