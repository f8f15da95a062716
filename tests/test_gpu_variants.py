"""Every shipped engine switch, exercised by the driver's ``-m gpu`` run (round 3 checked them with a shell script outside the
suite).  A switch selects another schedule / kernel for the SAME arithmetic up to summation order: three optimisation steps on
three configurations -- the headline layout on resident image batches with class weights, the BBOX-only input layer, a general
run shape (aggregate-first input layer, padded hidden width, output layer on the planes GEMMs) -- must give the default run's
losses to 1e-5 and its parameters to the conditioning of Adam's first steps.

Switches read by the Python engine are set in-process; switches the LIBRARY reads exist in the measurement build only
(libgte_hip_measure.so, -DGTE_MEASURE: the shipped library reads GTE_GEMM_MODE and nothing else) and are read once per process: the
same function runs in a child process that loads that build through GTE_LIB_PATH."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENGINE_SWITCHES = ["GTE_C_STEP=0", "GTE_P3_ROWS=0", "GTE_FUSE_ADAM=0", "GTE_TAIL_SPLIT=0", "GTE_FUSED_HEAD=0", "GTE_TRANSFORM_FIRST=0",
                   "GTE_PIPE_LATE=0", "GTE_FUSE_LN_FWD=0", "GTE_FUSE_LN_DX=0", "GTE_FUSE_LN_NARROW=0", "GTE_FUSE_SMALLK_DX=0",
                   "GTE_WIMG_IN_FOLD=0", "GTE_CACHE_AGG=0", "GTE_PLANES=0", "GTE_PLANES_GENERAL=0", "GTE_FUSE_HEAD_GEMM=0", "GTE_C_STEP=0 GTE_FUSE_LN_DX=0",
                   "GTE_C_STEP=0 GTE_FUSE_LN_NARROW=0", "GTE_PIPE_SIDE=1", "GTE_PIPE_SIDE=0", "GTE_PIPE_SIDE=1 GTE_PIPE_LATE=0", "GTE_PIPE_RIDE=0", "GTE_WIMG_BLOCK_MAJOR=0"]
LIBRARY_SWITCHES = ["GTE_SMALLK=0", "GTE_SMALLK_BWD=0", "GTE_NARROW_FWD16=0", "GTE_GEMM_MODE=f32", "GTE_P3_ROWS64=1", "GTE_P3_LN_ROWS=128",
                    "GTE_P3_LN_ROWS=64"]
CONFIGS = [(831, 256, True), (13, 256, False), (63, 200, False)]      # (F0, hidden, class weights)


def train_three_steps():
    """-> {config: (losses [3], flat parameters)} under the current environment"""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.data import synthetic as S
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    from gnn_tableextraction_amd.models.loop import BatchPipeline, run_steps
    dev = "cuda:0"
    out = {}
    for f0, hid, weighted in CONFIGS:
        pages = S.make_pages(14, in_feats=f0)
        graphs = []
        for p in pages:
            g = gte.PageGraph(p.src, p.dst, p.num_nodes)
            g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
            g.edata["feat"] = torch.from_numpy(p.weight)
            graphs.append(g)
        torch.manual_seed(3)
        model = gte.GcnSAGE(f0, hid, 9, 3, torch.nn.functional.relu, 0).to(dev)
        cw = torch.linspace(0.5, 2.0, 9, device=dev) if weighted else None
        tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4, class_weights=cw)
        res = G.ResidentPages(graphs, dev)
        pipe = BatchPipeline(res)          # (run_steps turns the resident features into images -- and, by default, caches the
                                           # input's mean aggregate next to them -- as train() and bench.py get it)
        losses = []
        steps = [np.array([(4 * s + j) % 14 for j in range(7)]) for s in range(3)]
        run_steps(tr, pipe, steps, on_step=lambda s, g, o: losses.append(o[:1].clone()))
        torch.cuda.synchronize()
        out[f"{f0}_{hid}"] = (np.array([float(l) for l in losses]), tr.flat_param.detach().cpu().numpy().copy())
    return out


_base = {}


def _baseline():
    if not _base:
        saved = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("GTE_") and k not in ("GTE_LIB_PATH",)}
        try:
            _base.update(train_three_steps())
        finally:
            os.environ.update(saved)
    return _base


def _compare(got, tag):
    base = _baseline()
    for k, (loss, params) in got.items():
        bl, bp = base[k]
        assert np.isfinite(loss).all(), f"{tag} {k}: {loss}"
        np.testing.assert_allclose(loss, bl, rtol=0, atol=2e-5, err_msg=f"{tag} {k}: losses")
        # three Adam steps of lr 0.01: an entry whose gradient is summation noise may move by up to 2 lr per step either way;
        # everything else follows the default run
        d = np.abs(params - bp)
        assert d.max() <= 0.061 and np.mean(d > 1e-4) < 0.03, f"{tag} {k}: parameters differ (max {d.max():.3e}, {np.mean(d > 1e-4):.4f} above 1e-4)"


@pytest.mark.parametrize("switch", ENGINE_SWITCHES)
def test_engine_switch_gives_the_default_runs_results(monkeypatch, switch):
    for kv in switch.split():
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    _compare(train_three_steps(), switch)


@pytest.mark.parametrize("switch", LIBRARY_SWITCHES)
def test_library_switch_gives_the_default_runs_results(tmp_path, switch):
    env = dict(os.environ)
    for kv in switch.split():
        k, v = kv.split("=")
        env[k] = v
    measure = os.path.join(ROOT, "gnn-tableextraction_amd", "libgte_hip_measure.so")
    assert os.path.exists(measure), "the measurement build is missing: make -C gnn-tableextraction_amd/csrc all"
    env["GTE_LIB_PATH"] = measure
    path = str(tmp_path / "out.npz")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from tests.test_gpu_variants import train_three_steps as t; r = t(); "
            "np.savez(%r, **{k + '.loss': v[0] for k, v in r.items()}, **{k + '.param': v[1] for k, v in r.items()})" % (ROOT, path))
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    z = np.load(path)
    got = {k[:-5]: (z[k], z[k[:-5] + ".param"]) for k in z.files if k.endswith(".loss")}
    _compare(got, switch)
