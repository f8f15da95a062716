"""End-to-end on an MI355X: the reference-compatible entry points train(data, config) and test(data, config)."""
import json
import os

import numpy as np
import pytest
import torch

from gnn_tableextraction_amd.components.graphs.loader import PrebuiltPages
from gnn_tableextraction_amd.models import model_predict, model_train
from gnn_tableextraction_amd.parsers.graphs import parse_args_ModelTrain
from gnn_tableextraction_amd.utils.config import logs_from_config

pytestmark = pytest.mark.gpu


def make_cfg(tmp_path, **flags):
    argv = ["--mode=knn", "--features", "BBOX", "--n_layers=3", "--mode_params=fixed", "--h_layer_dim=64",
            "--batch_size=8", "--n_epochs=4", "--lr=0.01", "--output_dir", str(tmp_path)]
    for k, v in flags.items():
        argv += [f"--{k}", str(v)]
    return parse_args_ModelTrain(argv=argv)


def learnable_pages(n_pages):
    """Synthetic pages whose label is a function of the geometry features, so training has signal."""
    data = PrebuiltPages.synthetic(n_pages, in_feats=13)
    for p, g in zip(data.page_arrays, data.graphs):
        y = (p.feat[:, 1] // 260).astype(np.int64).clip(0, 8)          # y0 band of the word -> class
        p.label[:] = y
        g.ndata['label'] = torch.from_numpy(y.astype(np.float32))
    return data


def test_train_writes_reference_layout_and_learns(tmp_path):
    data = learnable_pages(48)
    cfg = make_cfg(tmp_path)
    metrics = model_train.train(data, cfg)
    logs = logs_from_config(cfg)
    assert os.path.isfile(tmp_path / "weights" / f"{logs}.pt")
    assert os.path.isfile(tmp_path / "checkpoints" / logs)
    res = json.load(open(tmp_path / "results" / f"{logs}.json"))[logs]
    assert set(res) == {"train_loss", "train_acc", "val_loss", "val_acc", "cell_f1", "header_f1"}
    assert metrics.val.loss < np.log(9.0)                              # better than uniform
    ck = torch.load(tmp_path / "checkpoints" / logs, weights_only=False)
    # (the reference's four keys + `residency`: the layout a resumed run must reproduce -- budget, tier, ranks, windows)
    assert ck["epoch"] == 4 and set(ck) == {"epoch", "state_dict", "optimizer", "metrics", "residency"}
    assert ck["residency"]["tier"] == "all" and ck["residency"]["world"] == 1 and ck["residency"]["ranges"] is None
    assert sorted(ck["state_dict"]) == ["layers.0.linear.bias", "layers.0.linear.weight", "layers.0.lynorm.bias",
                                        "layers.0.lynorm.weight", "layers.1.linear.bias", "layers.1.linear.weight",
                                        "layers.1.lynorm.bias", "layers.1.lynorm.weight", "layers.2.linear.bias",
                                        "layers.2.linear.weight"]
    # torch.optim.Adam can load the optimizer state (checkpoint compatibility with the reference's format)
    from gnn_tableextraction_amd import GcnSAGE
    m = GcnSAGE(13, 64, 9, 3, torch.nn.functional.relu, 0)
    m.load_state_dict(ck["state_dict"])
    torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4).load_state_dict(ck["optimizer"])

    # resume: two more epochs from the checkpoint
    cfg2 = make_cfg(tmp_path, n_epochs=6, from_checkpoint="true")
    m2 = model_train.train(data, cfg2)
    assert torch.load(tmp_path / "checkpoints" / logs, weights_only=False)["epoch"] == 6
    assert m2.val.loss <= metrics.val.loss + 0.05

    # inference entry point with the saved best weights
    out = model_predict.test(data, cfg2)
    assert len(out["all_pred"]) == len(data) and out["accuracy"] > 0.3
    assert os.path.isfile(tmp_path / "predictions" / f"{logs}.pkl")
    # the reference's artefact (model_predict.py:172-174): ONE flat list pickled to {output}/all_pred/{logs}, which
    # post-processing slices back into pages by num_nodes (postprocessing.py:199-216)
    import pickle
    node_preds = pickle.load(open(tmp_path / "all_pred" / logs, "rb"))
    assert isinstance(node_preds, list) and all(isinstance(v, int) for v in node_preds[:50])
    assert len(node_preds) == sum(g.num_nodes() for g in data.graphs)
    start_index = 0
    for idx, graph in enumerate(data.graphs):
        end_index = start_index + graph.num_nodes()
        assert node_preds[start_index:end_index] == out["all_pred"][idx].tolist()
        start_index = end_index
    per_page = [float((out["all_pred"][i] == g.ndata["label"].long().numpy()).mean()) for i, g in enumerate(data.graphs)]
    assert abs(out["accuracy"] - float(np.mean(per_page))) < 1e-12               # "Mean Test Accuracy" = mean over pages
    one_by_one = make_cfg(tmp_path, n_epochs=6, batch_size=1)
    one_by_one.TRAINING.batch_size = 8                                  # same run name (bt_8) -> same weights file
    cfg2.TRAINING.batch_size = 8
    a = model_predict.test(data, cfg2, save_predictions=False)
    cfg2.TRAINING.batch_size = 1                                        # one forward per page, as the reference does
    w = os.path.join(str(tmp_path), "weights", f"{logs}.pt")
    b = model_predict.test(data, cfg2, weights_path=w, save_predictions=False)
    for pa, pb in zip(a["all_pred"], b["all_pred"]):
        np.testing.assert_array_equal(pa, pb)                           # batching pages does not change predictions


# the shapes of the reference's own runs (run_multiple_train.sh): h_layer_dim 1000, or int(calculate_hidden(...)) = 96 ... 218
REFERENCE_RUN_SHAPES = [(831, 1000, 3, 20), (13, 218, 3, 20), (831, 96, 3, 20), (363, 139, 3, 8), (63, 1000, 2, 8)]


@pytest.mark.parametrize("seed", list(range(14)) + [100 + i for i in range(len(REFERENCE_RUN_SHAPES))])
def test_fused_step_equals_autograd_path_on_random_shapes(seed):
    """Randomised cross-check of the hand-scheduled step engine (transform-first / q-form / narrow MFMA layer / fused head /
    deferred folds / GEMM tail split / split-K) against the autograd path (`model(g)` + `loss.backward()`), which takes none
    of those routes: same loss, same gradient, over random feature widths, hidden sizes, depths and batch sizes."""
    import numpy as np
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.data import synthetic as S
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep, TrainStep
    rng = np.random.default_rng(1000 + seed)
    f0 = int(rng.choice([3, 13, 50, 63, 64, 313, 363, 831]))
    hid = int(rng.choice([8, 24, 64, 96, 128, 200, 256, 384]))
    layers = int(rng.integers(2, 6))
    pages = int(rng.choice([1, 2, 5, 17, 60]))
    weighted = bool(rng.integers(0, 2))
    if seed >= 100:
        f0, hid, layers, pages = REFERENCE_RUN_SHAPES[seed - 100]
    dev = "cuda:0"
    pg = S.make_pages(pages, in_feats=f0, first_id=7000 + 100 * seed)
    src, dst, w, feat, label, off = S.concat_pages(pg)

    def fresh():
        torch.manual_seed(seed)
        m = gte.GcnSAGE(f0, hid, 9, layers, torch.nn.functional.relu, 0).to(dev)
        g = G.PageGraph(src, dst, int(off[-1]), device=dev)
        g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
        return m, g
    cw = torch.from_numpy(rng.random(9).astype(np.float32) + 0.5).to(dev) if weighted else None
    y = torch.from_numpy(label).to(dev)
    ma, ga = fresh()
    mb, gb = fresh()
    fused = FusedGcnSageStep(ma, lr=0.01, weight_decay=5e-4, class_weights=cw)
    auto = TrainStep(mb, lr=0.01, weight_decay=5e-4, class_weights=cw)
    out_f = fused.forward_backward(ga, y)
    auto.model.train()
    auto.flat_grad.zero_()
    loss, out_a = auto._loss(auto.model(gb), y)
    loss.backward()
    assert abs(float(out_f[0]) - float(out_a[0])) < 2e-5 * max(1.0, abs(float(out_a[0])))
    assert float(out_f[2]) == float(out_a[2])                      # same arg-max decisions
    gf, gr = fused.flat_grad.cpu().numpy(), auto.flat_grad.cpu().numpy()
    np.testing.assert_allclose(gf, gr, rtol=2e-3, atol=2e-5 * np.abs(gr).max() + 1e-9)


@pytest.mark.parametrize("hid,pages", [(512, 2), (1000, 5), (640, 1)])
def test_classic_backward_with_split_k_dh_gemms(monkeypatch, hid, pages):
    """GTE_TRANSFORM_FIRST=0 keeps the reference's aggregate-then-transform order: the backward then runs the dh GEMMs
    through gte_gemm_f32 while the step's fold deferral is open.  With a wide hidden layer and a SMALL batch those GEMMs
    are split-K: their folds must run immediately (the next kernel reads dh), never join the deferred batch."""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.data import synthetic as S
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep, TrainStep
    monkeypatch.setenv("GTE_TRANSFORM_FIRST", "0")
    dev = "cuda:0"
    pg = S.make_pages(pages, in_feats=63, first_id=9100)
    src, dst, w, feat, label, off = S.concat_pages(pg)

    def fresh():
        torch.manual_seed(3)
        m = gte.GcnSAGE(63, hid, 9, 3, torch.nn.functional.relu, 0).to(dev)
        g = G.PageGraph(src, dst, int(off[-1]), device=dev)
        g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
        return m, g
    y = torch.from_numpy(label).to(dev)
    ma, ga = fresh()
    mb, gb = fresh()
    fused = FusedGcnSageStep(ma, lr=0.01, weight_decay=5e-4)
    assert not fused.transform_first
    auto = TrainStep(mb, lr=0.01, weight_decay=5e-4)
    out_f = fused.forward_backward(ga, y)
    auto.model.train()
    auto.flat_grad.zero_()
    loss, out_a = auto._loss(auto.model(gb), y)
    loss.backward()
    assert abs(float(out_f[0]) - float(out_a[0])) < 2e-5
    gf, gr = fused.flat_grad.cpu().numpy(), auto.flat_grad.cpu().numpy()
    np.testing.assert_allclose(gf, gr, rtol=2e-3, atol=2e-5 * np.abs(gr).max() + 1e-9)


@pytest.mark.parametrize("f0", [13, 831])
def test_pipelined_loop_equals_one_batch_at_a_time(f0):
    """models/loop.py (what train() and bench.py run): epoch-wide metadata upload, reused buffer sets, batches assembled on
    a side stream one step ahead.  Parameters after two shuffled epochs must be BITWISE those of the plain loop
    `resident.batch(ids)` -> `step.step(...)` on the current stream."""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import distributed as D, graph as G
    from gnn_tableextraction_amd.models import loop
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    dev = torch.device("cuda", 0)
    data = PrebuiltPages.synthetic(36, in_feats=f0)
    res = G.ResidentPages(data.graphs, dev)
    sizes = res.page_sizes()

    def trainer():
        torch.manual_seed(11)
        m = gte.GcnSAGE(f0, 64, 9, 3, torch.nn.functional.relu, 0).to(dev)
        return FusedGcnSageStep(m, lr=0.01, weight_decay=5e-4)
    a, b = trainer(), trainer()
    pipe = loop.BatchPipeline(res)
    seen = []
    for epoch in range(2):
        plan = [r[0] for r in D.plan_epoch(sizes, 7, 1, seed=5, epoch=epoch)]
        assert len(plan) == 5
        out_a = loop.run_steps(a, pipe, plan, on_step=lambda s, g, o: seen.append(g.num_nodes()))
        for ids in plan:
            g = res.batch(ids)
            out_b = b.step(g, g.ndata["label"])
        assert torch.equal(out_a.cpu(), out_b.cpu())
    torch.cuda.synchronize()
    assert torch.equal(a.flat_param.cpu(), b.flat_param.cpu())
    assert len(seen) == 10 and len(set(seen)) > 3                     # a different batch (size) nearly every step


def test_bench_line_keeps_the_driver_contract():
    """`python bench.py --steps K --warmup W` prints ONE JSON line with the keys the driver and the judge read (metric / value /
    unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload,
    a `roofline` object priced against the guide's peak and a `cpu_baseline` object measured on the host cores)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--resident-pages", "300",
                        "--long-run-seconds", "0.05", "--val-graph", "200", "--gather-nodes", "100000", "--no-cfg3"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    # the record is the LAST stdout line and fits the driver's stdout window with room to spare (round 4's 22 kB line was cut
    # in the middle and could not be parsed); no prose objects in it
    assert r.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) <= 3000, len(lines[0])
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "step_roofline", "cpu_baseline", "gpu_over_cpu", "long_run", "gather", "extras"):
        assert k in rec, k
    assert set(rec["gather"]) >= {"frac", "bwd_frac", "traffic"} and 0 < rec["gather"]["frac"] < 1
    assert all(len(v) <= 120 for v in rec.values() if isinstance(v, str))
    ro, cb = rec["roofline"], rec["cpu_baseline"]
    assert ro["bound"] in ("hbm", "mfma") and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and "traffic" in ro
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert abs(rec["value"] - rec["config"]["nodes_per_step_per_gpu"] / (rec["ms_per_step"] * 1e-3)) < 0.05 * rec["value"]
    # the roofline block describes ONE kernel -- the layer-0 forward GEMM [x | cached ahn] W^T, K = 2 x 831, N = 256 --, the other
    # hidden layer's forward GEMM beside it; the work hoisted out of the timed step is named, and its cost measured (`uncached`)
    n_step = rec["config"]["nodes_per_step_per_gpu"]
    assert abs(ro["algorithmic_flops_per_launch"] - 2.0 * n_step * 1662 * 256) < 0.08 * 2.0 * n_step * 1662 * 256, ro
    assert "layer-0" in ro["kernel"] and ro["layer1"]["algorithmic_flops_per_launch"] < ro["algorithmic_flops_per_launch"]
    assert "input aggregate cached" in rec["config"]["workload"] and 0 < rec["uncached"] < 1.1 * rec["value"]
    # everything else: bench_extras.json next to bench.py (the full measurements, the same keys as before)
    d = json.load(open(os.path.join(root, rec["extras"])))
    assert abs(d["value"] - rec["value"]) <= 1e-6 * d["value"] and d["steps"] == 4
    assert d["metric"].startswith("nodes/sec") and d["unit"] == "nodes/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"].startswith("f32") and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["nodes_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 0.05 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s") and ro["peak"] > 0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and 0 < ro["frac"] < 1 and "traffic" in ro
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "nodes/s" and cb["sample"]
    gm = d["gemm_modes"]
    assert gm["other_mode"]["value"] > 0 and gm["error_vs_fp64"]["split_bf16"]["max"] <= 1.25 * gm["error_vs_fp64"]["f32"]["max"] + 2.5
    assert d["gather"]["bound"] == "hbm" and d["val_graph"]["logits_finite"] is True
