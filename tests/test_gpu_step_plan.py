"""gte_gcnsage_step (the whole optimisation step as one host call) against the call-by-call schedule of models/engine.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(monkeypatch, c_step, f0, hid, weighted, n_pages=12, steps=3, resident=False):
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.data import synthetic as S
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    monkeypatch.setenv("GTE_C_STEP", "1" if c_step else "0")
    dev = "cuda:0"
    pages = S.make_pages(n_pages, in_feats=f0)
    torch.manual_seed(7)
    model = gte.GcnSAGE(f0, hid, 9, 3, torch.nn.functional.relu, 0).to(dev)
    cw = torch.linspace(0.5, 2.0, 9, device=dev) if weighted else None
    tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4, class_weights=cw)
    assert tr.use_c_step == c_step
    if not tr._planes_on():
        pytest.skip("the one-call step drives the planes path (default GEMM mode, GTE_PLANES=1)")
    outs = []
    if resident:
        graphs = []
        for p in pages:
            g = gte.PageGraph(p.src, p.dst, p.num_nodes)
            g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
            g.edata["feat"] = torch.from_numpy(p.weight)
            graphs.append(g)
        res = G.ResidentPages(graphs, dev)
        if tr.wants_p3_features(f0):
            res.enable_p3()
    for s in range(steps):
        ids = [(3 * s + j) % n_pages for j in range(6)]
        if resident:
            g = res.batch(ids)
            y = g.ndata["label"]
        else:
            src, dst, w, feat, label, off = S.concat_pages([pages[i] for i in ids])
            g = G.PageGraph(src, dst, int(off[-1]), device=dev)
            g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
            y = torch.from_numpy(label).to(dev)
        outs.append(tr.step(g, y).cpu().numpy().copy())
    used = any(k for full in tr._bufs.values() for k in full.get("_plans", {}))
    if c_step and not used:
        pytest.skip("the engine's switches rule the one-call step out for this configuration")
    return (np.stack(outs), tr.flat_param.detach().cpu().numpy(), tr.flat_grad.detach().cpu().numpy(),
            tr.exp_avg_sq.detach().cpu().numpy(), used, res.p3_mode if resident else None)


@pytest.mark.parametrize("f0,hid,weighted,resident", [(831, 256, False, False), (831, 256, True, True), (13, 256, False, False),
                                                      (13, 256, True, True), (363, 128, False, False)])
def test_one_call_step_is_bitwise_the_call_by_call_step(monkeypatch, f0, hid, weighted, resident):
    """Same launches in the same order from C: losses, gradients, parameters and optimiser state after three steps are bit for
    bit those of the Python-issued schedule (planes input layer, BBOX-only input layer, class weights, resident batches with
    image features)."""
    a = _run(monkeypatch, True, f0, hid, weighted, resident=resident)
    b = _run(monkeypatch, False, f0, hid, weighted, resident=resident)
    assert a[4] and not b[4]                                     # the one-call path really ran / really did not
    for x, y in zip(a[:4], b[:4]):
        np.testing.assert_array_equal(x, y)
    assert np.isfinite(a[0]).all()


@pytest.mark.parametrize("c_step", [True, False])
def test_row_map_batches_are_bitwise_the_copied_image_batches(monkeypatch, c_step):
    """ResidentPages in image mode hands the input layer a ROW MAP into the resident image (GTE_P3_ROWS, default) instead of a
    copy of the batch's image rows: same bits after three steps, through the one-call step and the call-by-call schedule."""
    monkeypatch.setenv("GTE_P3_ROWS", "1")
    a = _run(monkeypatch, c_step, 831, 256, True, resident=True)
    monkeypatch.setenv("GTE_P3_ROWS", "0")
    b = _run(monkeypatch, c_step, 831, 256, True, resident=True)
    for x, y in zip(a[:4], b[:4]):
        np.testing.assert_array_equal(x, y)
    assert a[5] == "rows" and b[5] == "copy"


def test_step_plan_rejects_bad_plans():
    import ctypes
    from gnn_tableextraction_amd import _lib
    lib = _lib.load()
    plan = _lib.StepPlan()
    fused = ctypes.c_int(0)
    assert lib.gte_gcnsage_step(None, 0, ctypes.byref(fused), None) == -1
    assert lib.gte_gcnsage_step(ctypes.addressof(plan), 0, ctypes.byref(fused), None) == -1      # n_hidden = 0
    plan.n_hidden, plan.n_nodes = 1, 10
    assert lib.gte_gcnsage_step(ctypes.addressof(plan), 3, ctypes.byref(fused), None) == -1      # phase
    assert lib.gte_gcnsage_step(ctypes.addressof(plan), 0, ctypes.byref(fused), None) == -4      # no LayerNorm / bias: unsupported
    assert b"LayerNorm" in lib.gte_last_error()


@pytest.mark.parametrize("f0,hid,resident", [(831, 256, True), (831, 256, False), (13, 256, True), (363, 128, False)])
def test_forward_only_call_matches_the_module_forward_and_the_steps_forward(monkeypatch, f0, hid, resident):
    """gte_gcnsage_forward (engine.forward_logits): the logits of ``model(g)`` under no_grad (different summation order in the
    planes GEMMs: 2e-5 relative to the logit scale), and BIT FOR BIT the logits the training step's forward computes from the
    same weights; model_predict.predict_resident over all pages equals per-batch arg-max of those logits."""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G, _lib
    from gnn_tableextraction_amd.data import synthetic as S
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    from gnn_tableextraction_amd.models.loop import BatchPipeline
    from gnn_tableextraction_amd.models.model_predict import predict_resident
    dev = "cuda:0"
    pages = S.make_pages(10, in_feats=f0)
    torch.manual_seed(11)
    model = gte.GcnSAGE(f0, hid, 9, 3, torch.nn.functional.relu, 0).to(dev)
    tr = FusedGcnSageStep(model, lr=0.0, weight_decay=0.0)          # lr 0: a step leaves the weights as they are
    if not tr._planes_on():
        pytest.skip("the one-call plan drives the planes path (default GEMM mode, GTE_PLANES=1)")
    graphs = []
    for p in pages:
        g = gte.PageGraph(p.src, p.dst, p.num_nodes)
        g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat), torch.from_numpy(p.label.astype(np.float32))
        g.edata["feat"] = torch.from_numpy(p.weight)
        graphs.append(g)
    ids = [0, 3, 4, 7, 9]
    whole = G.batch([graphs[i] for i in ids]).to(dev)
    with torch.no_grad():
        want = model(whole).clone()
    if resident:
        res = G.ResidentPages(graphs, dev)
        if tr.wants_p3_features(f0):
            res.enable_p3()
        g = res.batch(ids)
        y = g.ndata["label"]
    else:
        g, y = whole, whole.ndata["label"]
    got = tr.forward_logits(g).clone()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1.0)
    if not any(k for full in tr._bufs.values() for k in full.get("_plans", {})):
        # (the module path answered, from the image's exact fp32 values for an image-mode batch)
        pytest.skip("the engine's switches rule the one-call plan out for this configuration")
    tr.step(g, y)
    n = want.shape[0]
    step_logits = next(iter(tr._bufs.values()))["y"][-1][:n]
    torch.testing.assert_close(step_logits, got, rtol=0, atol=0)
    if resident:
        pipe = BatchPipeline(res)
        pred = predict_resident(tr, pipe, 4)
        ref = []
        for b0 in range(0, len(graphs), 4):
            ref.append(tr.forward_logits(res.batch(list(range(b0, min(b0 + 4, len(graphs)))))).argmax(dim=1).clone())
        assert torch.equal(pred, torch.cat(ref))
        sub = predict_resident(tr, pipe, 3, page_ids=[5, 1, 2, 8])
        assert sub.shape[0] == sum(graphs[i].num_nodes() for i in (5, 1, 2, 8))
    lib = _lib.load()
    assert lib.gte_gcnsage_forward(None, None) == -1


@pytest.mark.parametrize("f0", [831, 13])
def test_fold_launch_writes_the_weight_images_of_the_updated_parameters(monkeypatch, f0):
    """gte_fold_defer_flush_adam_images: after a one-call step the weight images are, byte for byte, what the conversion launch
    makes of the updated parameters (so the next forward skips it); a parameter changed through torch is noticed (version
    counters) and the images are converted again; GTE_WIMG_IN_FOLD=0 gives the same bits."""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G
    from gnn_tableextraction_amd.data import synthetic as S
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    dev = "cuda:0"
    pages = S.make_pages(8, in_feats=f0)
    src, dst, w, feat, label, off = S.concat_pages(pages[:5])
    g = G.PageGraph(src, dst, int(off[-1]), device=dev)
    g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).to(dev), torch.from_numpy(w).to(dev)
    y = torch.from_numpy(label).to(dev)
    finals = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GTE_WIMG_IN_FOLD", flag)
        torch.manual_seed(3)
        model = gte.GcnSAGE(f0, 256, 9, 3, torch.nn.functional.relu, 0).to(dev)
        tr = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
        if not tr._planes_on():
            pytest.skip("the one-call step drives the planes path (default GEMM mode, GTE_PLANES=1)")
        for _ in range(3):
            tr.step(g, y)
        if not any(k for full in tr._bufs.values() for k in full.get("_plans", {})):
            pytest.skip("the engine's switches rule the one-call step out for this configuration")
        if not tr.fuse_adam:
            pytest.skip("GTE_FUSE_ADAM=0: the optimiser step is its own launch")
        if flag == "1":
            assert tr.adam_fused_steps == 3 and tr._wimg_sig is not None          # the fold launch wrote them
            have = {i: (fw.data.clone(), None if bw is None else bw.data.clone()) for i, (fw, bw) in tr._wimg.items()}
            assert have
            dims = [f0] + [l.out_feats for l in model.layers]
            tr._weight_images(dims, launch=True)                                    # the conversion launch on the same parameters
            for i, (fw, bw) in tr._wimg.items():
                assert torch.equal(have[i][0], fw.data)
                if bw is not None:
                    assert torch.equal(have[i][1], bw.data)
            # a change through torch is noticed: the forward-only call converts again and matches the module path
            with torch.no_grad():
                model.layers[1].linear.weight.mul_(0.5)
                want = model(g).clone()
            got = tr.forward_logits(g)
            assert float((got - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1.0)
            with torch.no_grad():
                model.layers[1].linear.weight.mul_(2.0)                             # exact: back to the trained weights
        else:
            assert tr._wimg_sig is None
        tr.step(g, y)
        finals.append(tr.flat_param.detach().cpu().numpy().copy())
    np.testing.assert_array_equal(finals[0], finals[1])


def test_fold_flush_with_images_rejects_bad_image_lists():
    """gte_fold_defer_flush_adam_images: an image that is not a sub-matrix of the parameter buffer, or a negative count, is an error
    (the deferral is closed either way); without an open deferral the call fails; more than 8 images are not an error (the step
    is applied, the images are left to the conversion launch: *fused = 1 at most)."""
    import ctypes
    from gnn_tableextraction_amd import _lib, ops
    lib, P = _lib.load(), _lib.ptr
    dev = "cuda:0"
    n = 4096
    param, grad, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    hyper = torch.tensor([0.01, 0.9, 0.999, 1e-8, 0.0, 1.0, 0.1, 0.0316], device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    ticket = torch.zeros(int(lib.gte_adam_ticket_bytes()) // 4, dtype=torch.int32, device=dev)
    other = torch.zeros(64, 64, device=dev)
    img = ops.P3.empty(64, 64, dev)
    fused = ctypes.c_int(7)
    args = (P(param), P(grad), P(m), P(v), n, P(hyper), P(step), P(ticket))
    assert lib.gte_fold_defer_flush_adam_images(*args, None, 0, ctypes.byref(fused)) == -1          # no deferral open
    bad = (_lib.P3Desc * 1)(_lib.P3Desc(other.data_ptr(), 64, 64, 64, 0, img.data.data_ptr(), img.ldp))   # not inside `param`
    assert lib.gte_fold_defer_begin(_lib.current_stream()) == 0
    assert lib.gte_fold_defer_flush_adam_images(*args, ctypes.addressof(bad), 1, ctypes.byref(fused)) == -1
    assert b"sub-matrix" in lib.gte_last_error() and fused.value == 0
    assert lib.gte_fold_defer_begin(_lib.current_stream()) == 0                                     # (the failed call closed it)
    assert lib.gte_fold_defer_flush_adam_images(*args, None, -1, ctypes.byref(fused)) == -1
    good = _lib.P3Desc(param.data_ptr(), 64, 64, 64, 0, img.data.data_ptr(), img.ldp)
    many = (_lib.P3Desc * 9)(*([good] * 9))
    assert lib.gte_fold_defer_begin(_lib.current_stream()) == 0
    assert lib.gte_fold_defer_flush_adam_images(*args, ctypes.addressof(many), 9, ctypes.byref(fused)) == 0
    assert fused.value == 0                                                                          # nothing queued: no step either
    torch.cuda.synchronize()
