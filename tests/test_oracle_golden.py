"""The CPU oracle (oracle/gcnsage_cpu.py) against the golden vectors produced by the
reference's own models.py under a stub dgl (oracle/make_golden.py), and against an
independent fp64 dense-adjacency formulation.  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import gcnsage_cpu as oc
from tests import poststep
from tests.conftest import GOLDEN_DIR

GCN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                   if not os.path.basename(p).startswith(("meansage", "aux_", "headline", "shape_")))
SHAPE_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "shape_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n = int(z["meta"][0])
    g = oc.OracleGraph(z["src"], z["dst"], n, z["w"])
    state0 = {k[len("state0."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state0.")}
    return z, g, state0


def test_golden_present():
    assert len(GCN_CASES) >= 8 and "tiny_6n_10e" in GCN_CASES


@pytest.mark.parametrize("name", GCN_CASES)
def test_forward_matches_reference(name):
    z, g, state0 = load_case(name)
    logits, hidden = oc.gcnsage_forward(state0, g, torch.from_numpy(z["x"]), return_hidden=True)
    # tolerance of BASELINE.json north_star: 1e-5 fp32 on the forward (logits)
    np.testing.assert_allclose(logits.numpy(), z["logits"], atol=1e-5, rtol=1e-5)
    for i, h in enumerate(hidden):
        np.testing.assert_allclose(h.numpy(), z[f"hidden.{i}"], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name", GCN_CASES)
def test_train_step_matches_reference(name):
    z, g, state0 = load_case(name)
    cw = torch.from_numpy(z["class_weights"]) if "class_weights" in z.files else None
    tr = oc.OracleTrainer(state0, lr=0.01, weight_decay=5e-4, class_weights=cw)
    loss, _ = tr.step(g, torch.from_numpy(z["x"]), torch.from_numpy(z["y"]))
    assert abs(loss - float(z["loss"])) < 1e-5
    for k, gr in tr.grads().items():
        ref = z["grad." + k]
        np.testing.assert_allclose(gr.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=1e-4)
    # resolved gradients -> parameters equal to 1e-5; post-step logits at 1e-4 (tests/poststep.py explains the mask)
    with torch.no_grad():
        after = oc.gcnsage_forward({k: v.detach() for k, v in tr.state.items()}, g, torch.from_numpy(z["x"]))
    poststep.check_against_fixture(z, {k: v.detach().numpy() for k, v in tr.state.items()}, after.numpy())


@pytest.mark.parametrize("name", [c for c in GCN_CASES if c not in ("page300_f831_l3",)])
def test_dense_fp64_cross_check(name):
    z, g, state0 = load_case(name)
    n = int(z["meta"][0])
    dense = oc.dense_reference_forward({k: v.numpy() for k, v in state0.items()},
                                       z["src"], z["dst"], z["w"], n, z["x"])
    np.testing.assert_allclose(dense, z["logits"], atol=1e-5, rtol=1e-5)


def test_meansage_matches_reference():
    z = np.load(os.path.join(GOLDEN_DIR, "meansage_120.npz"))
    g = oc.OracleGraph(z["src"], z["dst"], int(z["meta"][0]), z["w"])
    n_lin = len([k for k in z.files if k.endswith("linear.weight")])
    ws = [(torch.from_numpy(z[f"state0.layers.{i}.linear.weight"]),
           torch.from_numpy(z[f"state0.layers.{i}.linear.bias"])) for i in range(n_lin)]
    out = oc.meansage_forward(ws, g, torch.from_numpy(z["x"]))
    np.testing.assert_allclose(out.numpy(), z["out"], atol=1e-5, rtol=1e-5)


def test_hand_computed_tiny_aggregation():
    """6 nodes / 10 edges: node 2 has in-edges from 1 (w .5), 0 (w .5, twice: a duplicate
    edge), 2 (self loop, w 1); node 5 has none."""
    z, g, _ = load_case("tiny_6n_10e")
    x = np.arange(18, dtype=np.float32).reshape(6, 3)
    ah = oc.spmm_csr_numpy(g.indptr, g.indices, g.weight, x)
    np.testing.assert_allclose(ah[2], 0.5 * x[1] + 0.5 * x[0] + 0.5 * x[0] + 1.0 * x[2])
    np.testing.assert_allclose(ah[5], 0.0)
    assert g.norm[5, 0] == 0.0 and g.norm[2, 0] == np.float32(0.25)
    t = oc.spmm_csr_torch(g.indptr, g.indices, g.weight, torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(t, ah, rtol=1e-6)


def test_omp_kernel_matches_numpy_when_built():
    if not oc.omp_available():
        pytest.skip("oracle/_build/liboracle_spmm.so not built")
    rng = np.random.default_rng(0)
    n, e, f = 500, 4000, 37
    g = oc.OracleGraph(rng.integers(0, n, e), rng.integers(0, n, e), n, rng.uniform(0, 1, e))
    x = rng.standard_normal((n, f)).astype(np.float32)
    a = oc.spmm_csr_torch(g.indptr, g.indices, g.weight, torch.from_numpy(x)).numpy()
    b = oc.spmm_csr_numpy(g.indptr, g.indices, g.weight, x)
    np.testing.assert_array_equal(a, b)      # same sequential order, contraction off -> bit-equal


def test_shape_helpers_known_answers():
    # components/features/utils.py:71-101 ; values from SURVEY 8(c)(4), hand-computable
    assert oc.get_in_feats(["BBOX"]) == 13
    assert oc.get_in_feats(["BBOX", "REPR", "SCIBERT"]) == 831
    assert oc.get_in_feats(["SPACY"], padding=True) == 831
    assert int(oc.calculate_hidden(13, 9, 100000, 3)) == 218
    assert int(oc.calculate_hidden(831, 9, 100000, 3)) == 96
    h = oc.calculate_hidden(10000, 8, 100000, 3)      # the module's own __main__ example
    assert abs(2 * h * h + 10008 * h - 100000) < 1e-6


def test_headline_shape_case_matches_reference():
    """(2 000 nodes, F0 = 831, H = 256, 3 layers) -- the headline model; trimmed fixture, inputs and initial
    weights regenerated from the seed."""
    z, src, dst, w, x, y, state0, _ = poststep.headline_case(GOLDEN_DIR)
    g = oc.OracleGraph(src, dst, len(x), w)
    logits, hidden = oc.gcnsage_forward(state0, g, torch.from_numpy(x), return_hidden=True)
    tr = oc.OracleTrainer(state0, lr=0.01, weight_decay=5e-4)
    loss, _ = tr.step(g, torch.from_numpy(x), torch.from_numpy(y))
    with torch.no_grad():
        after = oc.gcnsage_forward({k: v.detach() for k, v in tr.state.items()}, g, torch.from_numpy(x)).numpy()
    poststep.check_headline(z, logits.numpy(), [h.numpy() for h in hidden], loss, {k: v.numpy() for k, v in tr.grads().items()},
                            {k: v.detach().numpy() for k, v in tr.state.items()}, after, state0)


@pytest.mark.parametrize("name", SHAPE_CASES)
def test_reference_run_shapes_match_reference(name):
    """The reference's OWN run shapes (run_multiple_train.sh:8-113: --h_layer_dim=1000, or int(calculate_hidden) for 100 000
    parameters): hidden 1000 / 218 / 149 / 139 / 100 / 96 with F0 = 13 ... 831, trimmed fixtures generated from the reference's
    models.py; the oracle's forward, loss, gradients and post-step state against them."""
    z, src, dst, w, x, y, state0, _ = poststep.trimmed_case(GOLDEN_DIR, name)
    g = oc.OracleGraph(src, dst, len(x), w)
    logits, hidden = oc.gcnsage_forward(state0, g, torch.from_numpy(x), return_hidden=True)
    tr = oc.OracleTrainer(state0, lr=0.01, weight_decay=5e-4)
    loss, _ = tr.step(g, torch.from_numpy(x), torch.from_numpy(y))
    with torch.no_grad():
        after = oc.gcnsage_forward({k: v.detach() for k, v in tr.state.items()}, g, torch.from_numpy(x)).numpy()
    poststep.check_headline(z, logits.numpy(), [h.numpy() for h in hidden], loss, {k: v.numpy() for k, v in tr.grads().items()},
                            {k: v.detach().numpy() for k, v in tr.state.items()}, after, state0)


def test_shape_cases_cover_the_reference_grid():
    assert {"shape_f13_h218", "shape_f363_h149", "shape_f63_h1000", "shape_f831_h96", "shape_f831_h1000"} <= set(SHAPE_CASES)
    # every scaled width of run_multiple_train.sh (int(calculate_hidden) for F0 = 13 / 63 / 313 / 363 / 781 / 831)
    assert {"shape_f13_h218", "shape_f63_h206", "shape_f313_h157", "shape_f363_h149", "shape_f781_h100", "shape_f831_h96"} <= set(SHAPE_CASES)
