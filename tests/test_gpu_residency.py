"""What the windowed residency changes is the ORDER of the samples (models/residency.py: a window is visited for
``GTE_WINDOW_PASSES`` shuffled passes before the stream moves on), not the arithmetic (bitwise on the same step stream:
tests/test_gpu_shapes.py).  The reference reshuffles ALL training pages every epoch (src/models/model_train.py:279-283).  This
file measures what the changed order costs in validation loss after a fixed number of steps, against the reference's order."""
import json
import os

import numpy as np
import pytest
import torch

import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import distributed as D, graph as G, ops
from gnn_tableextraction_amd.data import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _learnable_pages(n_pages, f0=13):
    """Synthetic pages with labels that are a function of the BBOX geometry features (vertical band + horizontal third of the
    word), sorted by page size: contiguous windows then differ in their page-size (and so label) statistics -- the hard case for
    a loop that trains on one window for several passes."""
    pages = S.make_pages(n_pages, in_feats=f0)
    pages.sort(key=lambda p: p.num_nodes)
    graphs = []
    for p in pages:
        y = ((p.feat[:, 1] // 300).astype(np.int64) + 3 * (p.feat[:, 0] // 560).astype(np.int64)).clip(0, 8)
        g = gte.PageGraph(p.src, p.dst, p.num_nodes)
        g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(y.astype(np.float32))
        g.edata["feat"] = torch.from_numpy(p.weight)
        graphs.append(g)
    return graphs


def test_windowed_order_reaches_the_validation_loss_of_the_reference_order():
    """3 000 training pages (F0 = 13, hidden 64), 20 pages per step, 1 500 steps = 10 epochs' worth, three orders:
    (a) all-resident, every epoch a fresh shuffle of all pages (distributed.plan_epoch = model_train.py:279-283);
    (b) windowed, 1 pass per window visit (every page once per sweep of the windows: the closest a windowed loop gets to (a));
    (c) windowed, 8 passes per visit (the default of train() under a budget);
    (d) as (c) with the pages laid out sorted by size instead of train()'s seeded random order -- reported only.
    Same initial weights, same optimiser, same 300 held-out validation pages in one graph.  Asserted: the windowed orders end
    no more than 10 % above (a)'s validation loss and all three learn (loss far below ln 9).  The measured losses are written to
    gpurun_out/residency_semantics.json for DESIGN.md."""
    from gnn_tableextraction_amd.models import residency as R
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    from gnn_tableextraction_amd.models.loop import BatchPipeline, run_steps
    from gnn_tableextraction_amd.models.model_train import evaluate
    f0, hid, B, n_steps = 13, 64, 20, 1500
    graphs = _learnable_pages(3300, f0)
    rng = np.random.default_rng(3)
    held = np.zeros(len(graphs), dtype=bool)
    held[rng.choice(len(graphs), 300, replace=False)] = True
    train = [g for g, h in zip(graphs, held) if not h]
    val = G.batch([g.to(DEV) for g, h in zip(graphs, held) if h])
    val_y = val.ndata["label"]

    def fresh():
        torch.manual_seed(21)
        m = gte.GcnSAGE(f0, hid, 9, 3, torch.nn.functional.relu, 0).to(DEV)
        return m, FusedGcnSageStep(m, lr=0.01, weight_decay=5e-4)

    out = {}
    # (a) the reference's order
    m, tr = fresh()
    res = G.ResidentPages(train, DEV)
    pipe = BatchPipeline(res)
    sizes = res.page_sizes()
    done, epoch = 0, 0
    while done < n_steps:
        plan = [r[0] for r in D.plan_epoch(sizes, B, 1, seed=42, epoch=epoch)][:n_steps - done]
        run_steps(tr, pipe, plan)
        done += len(plan)
        epoch += 1
    out["all_resident"] = evaluate(m, val, val_y, engine=tr)[:2]
    del pipe, res
    # (b), (c) windows of ~1/8 of the set, the pages laid out in the seeded random order train() uses (window_page_order);
    # (d) the same with the pages left sorted by size: windows that are biased samples of the set (reported, not asserted --
    # the reason train() shuffles the layout)
    order = R.window_page_order(len(train), 42, 0)
    shuffled = [train[i] for i in order]
    for tag, pages_, passes in (("windowed_passes_1", shuffled, 1), ("windowed_passes_8", shuffled, 8), ("sorted_windows_passes_8", train, 8)):
        host = R.HostPages(pages_, DEV)
        m, tr = fresh()
        want_p3 = tr.wants_p3_features(f0)
        per_node = R.WindowedPages.bytes_per_node(host.page_nodes, host.page_edges, f0, want_p3)
        wp = R.WindowedPages(host, float(host.page_nodes.sum()) * per_node / 4.0, want_p3)
        assert len(wp.ranges) >= 6
        stream = R.WindowStream(wp.ranges, B, passes, 42)
        wp.prefetch(stream.peek_window())
        pipe = BatchPipeline(wp.acquire(stream.peek_window()))
        pipe._bound_pages = (host.page_nodes, np.diff(host.sets["in"]["edge_off"]), np.diff(host.sets["out"]["edge_off"]))
        R.run_windowed(tr, pipe, wp, stream, n_steps)
        torch.cuda.synchronize()
        out[tag] = evaluate(m, val, val_y, engine=tr)[:2] + (len(wp.ranges),)
        del pipe, wp, host
    ref = out["all_resident"][0]
    report = {"workload": f"{len(train)} training pages sorted by size (F0 = {f0}, hidden {hid}), {B} pages per step, {n_steps} steps; "
                          f"300 held-out pages", "val_loss": {k: v[0] for k, v in out.items()}, "val_acc": {k: v[1] for k, v in out.items()},
              "windows": out["windowed_passes_8"][2], "rel_to_all_resident": {k: v[0] / ref for k, v in out.items()}}
    print("residency semantics:", json.dumps(report))
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        json.dump(report, open(os.path.join(d, "residency_semantics.json"), "w"), indent=1)
    assert all(v[0] < 0.6 * np.log(9.0) for k, v in out.items() if not k.startswith("sorted")), report
    for k in ("windowed_passes_1", "windowed_passes_8"):       # (one-sided: measured 0.294 - 0.297 against 0.337 -- run-to-run spread)
        assert out[k][0] <= 1.10 * ref + 0.01, report
