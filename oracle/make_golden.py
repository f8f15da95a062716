"""Generate golden vectors from the REFERENCE's own model code  (build container only).

Runs ``/root/reference/src/components/graphs/models.py`` (GcnSAGE / MeanSAGE) under
a stub ``dgl`` module and stores inputs + outputs as ``tests/golden/*.npz``.
Nothing from the reference is copied: the fixtures are data (inputs, the
reference's ``state_dict`` values, activations, loss, gradients, one Adam step).

The stub encodes DGL's documented gSpMM semantics for the two call sites the
model uses (models.py:53-54 and :146-149): message ``h[u] * w_e`` (scalar edge
weight broadcast over features), reducer ``sum`` / ``mean`` over the incoming
edges of each destination, zeros for in-degree-0 nodes.  It deliberately uses a
different algorithm (gather + index_add_) from the oracle's CSR SpMM.

Usage:  python oracle/make_golden.py            (needs /root/reference; never runs
on the GPU box -- the committed .npz files travel instead).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF_MODELS = "/root/reference/src/components/graphs/models.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# ------------------------------ stub dgl ----------------------------------- #
def _install_stub_dgl():
    dgl = types.ModuleType("dgl")
    nn_ = types.ModuleType("dgl.nn")
    pt = types.ModuleType("dgl.nn.pytorch")
    conv = types.ModuleType("dgl.nn.pytorch.conv")
    conv.SAGEConv = object
    fn = types.ModuleType("dgl.function")
    fn.u_mul_e = lambda u, e, m: ("u_mul_e", u, e, m)
    fn.copy_u = lambda u, m: ("copy_u", u, m)
    fn.sum = lambda msg, out: ("sum", msg, out)
    fn.mean = lambda msg, out: ("mean", msg, out)
    dgl.nn, nn_.pytorch, pt.conv, dgl.function = nn_, pt, conv, fn
    for name, mod in [("dgl", dgl), ("dgl.nn", nn_), ("dgl.nn.pytorch", pt),
                      ("dgl.nn.pytorch.conv", conv), ("dgl.function", fn)]:
        sys.modules[name] = mod


class StubGraph:
    def __init__(self, src, dst, n):
        self.src = torch.as_tensor(src, dtype=torch.long)
        self.dst = torch.as_tensor(dst, dtype=torch.long)
        self.n = n
        self.ndata, self.edata = {}, {}

    def local_var(self):
        g = StubGraph(self.src, self.dst, self.n)
        g.ndata, g.edata = dict(self.ndata), dict(self.edata)
        return g

    class _Scope:
        def __init__(self, g):
            self.g = g

        def __enter__(self):
            self.nd, self.ed = dict(self.g.ndata), dict(self.g.edata)

        def __exit__(self, *a):
            self.g.ndata, self.g.edata = self.nd, self.ed

    def local_scope(self):
        return StubGraph._Scope(self)

    def in_degrees(self):
        return torch.bincount(self.dst, minlength=self.n)

    def update_all(self, message_func=None, reduce_func=None):
        mf, rf = message_func, reduce_func
        x = self.ndata[mf[1]][self.src]
        if mf[0] == "u_mul_e":
            x = x * self.edata[mf[2]].unsqueeze(-1)
        out = torch.zeros(self.n, x.shape[1], dtype=x.dtype).index_add_(0, self.dst, x)
        if rf[0] == "mean":
            out = out / self.in_degrees().clamp(min=1).to(x.dtype)[:, None]
        self.ndata[rf[2]] = out


def _load_reference_models():
    _install_stub_dgl()
    spec = importlib.util.spec_from_file_location("ref_models", REF_MODELS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------ graphs -------------------------------------- #
def knn_like_graph(rng, n, k, bidirect=True):
    """Page-like graph: nodes on a jittered reading-order grid, k nearest by
    centre distance, neighbour -> node, optional to_simple + to_bidirected."""
    cols = max(2, int(np.sqrt(n)))
    pos = np.stack([(np.arange(n) % cols) * 60.0, (np.arange(n) // cols) * 25.0], 1)
    pos += rng.uniform(-8, 8, size=pos.shape)
    d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    np.fill_diagonal(d, np.inf)
    kk = min(k, n - 1)
    nbr = np.argsort(d, axis=1)[:, :kk]
    src = nbr.reshape(-1)
    dst = np.repeat(np.arange(n), kk)
    if bidirect:
        pairs = set(zip(src.tolist(), dst.tolist()))
        pairs |= {(b, a) for a, b in pairs}
        pairs = sorted(pairs, key=lambda p: (p[1], p[0]))
        src = np.array([p[0] for p in pairs], dtype=np.int64)
        dst = np.array([p[1] for p in pairs], dtype=np.int64)
    dist = np.floor(d[dst, src])
    m = dist.max() if len(dist) and dist.max() > 0 else 1.0
    w = (1.0 - dist / m).astype(np.float32)                      # loader.py:341-344
    return src.astype(np.int64), dst.astype(np.int64), w


def random_graph(rng, n, e):
    return rng.integers(0, n, e), rng.integers(0, n, e), rng.uniform(0, 1, e).astype(np.float32)


# ------------------------------ cases --------------------------------------- #
def run_gcnsage_case(ref, name, src, dst, w, n, f0, hid, n_cls, n_layers, seed, x_scale=1.0,
                     class_weights=None, bbox_like=False):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    model = ref.GcnSAGE(f0, hid, n_cls, n_layers, F.relu, 0)
    if bbox_like:   # raw-pixel magnitudes like nlp/bbox.py geometry features
        x = np.concatenate([rng.uniform(0, 2000, (n, 6)), rng.uniform(0, 5e5, (n, 3)),
                            rng.uniform(0, 1, (n, f0 - 9))], 1).astype(np.float32)
    else:
        x = (rng.standard_normal((n, f0)) * x_scale).astype(np.float32)
    y = rng.integers(0, n_cls, n).astype(np.int64)
    g = StubGraph(src, dst, n)
    g.ndata["feat"] = torch.from_numpy(x)
    g.edata["feat"] = torch.from_numpy(np.asarray(w, dtype=np.float32))
    state0 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}

    hidden = []
    hooks = [l.register_forward_hook(lambda m, i, o: hidden.append(o.detach().numpy().copy()))
             for l in model.layers]
    cw = None if class_weights is None else torch.tensor(class_weights, dtype=torch.float32)
    loss_fn = torch.nn.CrossEntropyLoss(weight=cw)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)   # model_train.py:168
    logits = model(g)
    for h in hooks:
        h.remove()
    loss = loss_fn(logits, torch.from_numpy(y))
    opt.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone().numpy() for k, p in model.named_parameters()}
    opt.step()
    state1 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    with torch.no_grad():
        logits1 = model(g).numpy()

    out = dict(src=src, dst=dst, w=np.asarray(w, dtype=np.float32), x=x, y=y,
               meta=np.array([n, f0, hid, n_cls, n_layers, seed], dtype=np.int64),
               logits=logits.detach().numpy(), loss=np.float32(loss.item()), logits_after_step=logits1)
    if class_weights is not None:
        out["class_weights"] = np.asarray(class_weights, dtype=np.float32)
    for i, h in enumerate(hidden):
        out[f"hidden.{i}"] = h
    for k, v in state0.items():
        out["state0." + k] = v
    for k, v in grads.items():
        out["grad." + k] = v
    for k, v in state1.items():
        out["state1." + k] = v
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={n} E={len(src)} F0={f0} H={hid} L={n_layers} loss={loss.item():.6f} "
          f"-> {os.path.getsize(path) / 1e3:.0f} kB")


def run_meansage_case(ref, name, src, dst, w, n, f0, hid, n_cls, n_layers, seed):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    model = ref.MeanSAGE(f0, hid, n_cls, n_layers)
    x = rng.standard_normal((n, f0)).astype(np.float32)
    g = StubGraph(src, dst, n)
    out_t = model(g, torch.from_numpy(x), torch.from_numpy(np.asarray(w, dtype=np.float32)))
    out = dict(src=src, dst=dst, w=np.asarray(w, dtype=np.float32), x=x,
               meta=np.array([n, f0, hid, n_cls, n_layers, seed], dtype=np.int64),
               out=out_t.detach().numpy())
    for k, v in model.state_dict().items():
        out["state0." + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT_DIR, name + ".npz"), **out)
    print(f"{name}: MeanSAGE N={n} E={len(src)}")


def headline_inputs(seed=11, n=2000, f0=831):
    """Inputs of the headline-shape case, regenerated from the seed by the tests (not stored: 6.6 MB of N(0,1))."""
    rng = np.random.default_rng(seed)
    s, d, ww = knn_like_graph(rng, n, 5)
    x = rng.standard_normal((n, f0)).astype(np.float32)
    y = rng.integers(0, 9, n).astype(np.int64)
    return s, d, ww, x, y


def shape_inputs(seed, n, f0, bbox_like=False):
    """Inputs of a trimmed run-shape case (run_shape_case), regenerated from the seed by the tests.  ``bbox_like``: the first
    nine features carry raw-pixel magnitudes like the geometry features of nlp/bbox.py:49-54 (un-normalised, up to 5e5)."""
    rng = np.random.default_rng(seed)
    s, d, ww = knn_like_graph(rng, n, 5)
    if bbox_like:
        x = np.concatenate([rng.uniform(0, 2000, (n, 6)), rng.uniform(0, 5e5, (n, 3)), rng.uniform(0, 1, (n, f0 - 9))], 1).astype(np.float32)
    else:
        x = rng.standard_normal((n, f0)).astype(np.float32)
    y = rng.integers(0, 9, n).astype(np.int64)
    return s, d, ww, x, y


# The reference's OWN run shapes (run_multiple_train.sh:8-113 with model_train.py:81-91,157 and components/features/utils.py:90-101):
# --h_layer_dim=1000, or --mode_params=scaled --params_no=100000 -> int(calculate_hidden) = 218 / 206 / 157 / 149 / 100 / 96 for
# F0 = 13 / 63 / 313 / 363 / 781 / 831.  (name, seed, nodes, F0, hidden, bbox-like inputs)
SHAPE_CASES = [("shape_f13_h218", 21, 640, 13, 218, True), ("shape_f363_h149", 22, 512, 363, 149, False),
               ("shape_f363_h139", 23, 384, 363, 139, False), ("shape_f63_h1000", 24, 400, 63, 1000, False),
               ("shape_f831_h96", 25, 512, 831, 96, False), ("shape_f831_h1000", 26, 320, 831, 1000, False),
               ("shape_f13_h1000", 27, 320, 13, 1000, True), ("shape_f781_h100", 28, 384, 781, 100, False),
               ("shape_f63_h206", 29, 448, 63, 206, False), ("shape_f313_h157", 30, 448, 313, 157, False)]


def run_shape_case(ref, name, seed, n, f0, hid, bbox_like=False):
    """One of the reference's run shapes, GcnSAGE(f0, hid, 9, 3), in the TRIMMED format of the headline case (inputs and initial
    weights are regenerated from the seed; the fixture holds the graph, logits, loss, post-step logits, small gradients whole,
    8 192 sampled entries + sum + max of every large tensor, every 16th row of the hidden activations)."""
    s, d, ww, x, y = shape_inputs(seed, n, f0, bbox_like)
    _run_trimmed(ref, name, seed, n, f0, hid, s, d, ww, x, y, extra_meta=[int(bbox_like)])


def run_headline_case(ref, name="headline_n2000_f831_h256"):
    """(N, E, F0, H, L) = (2 000, ~16 000, 831, 256, 3) -- the headline model on a 2 000-node graph.  Stored
    TRIMMED (the full case is ~17 MB): graph, logits, loss, post-step logits, every small gradient, 8 192 sampled entries + sum
    + max of every large tensor, every 16th row of the hidden activations.  Inputs are regenerated from the seed
    (``headline_inputs``), the initial weights by the seed through the model constructor (the repository reproduces the
    reference's RNG stream bit for bit; the fixture carries sums to check that)."""
    seed, n, f0, hid = 11, 2000, 831, 256
    s, d, ww, x, y = headline_inputs(seed, n, f0)
    _run_trimmed(ref, name, seed, n, f0, hid, s, d, ww, x, y)


def _run_trimmed(ref, name, seed, n, f0, hid, s, d, ww, x, y, extra_meta=()):
    torch.manual_seed(seed)
    model = ref.GcnSAGE(f0, hid, 9, 3, F.relu, 0)
    g = StubGraph(s, d, n)
    g.ndata["feat"], g.edata["feat"] = torch.from_numpy(x), torch.from_numpy(ww)
    state0 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    hidden = []
    hooks = [l.register_forward_hook(lambda m, i, o: hidden.append(o.detach().numpy().copy())) for l in model.layers]
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    logits = model(g)
    for h in hooks:
        h.remove()
    loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(y))
    opt.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone().numpy() for k, p in model.named_parameters()}
    opt.step()
    state1 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    with torch.no_grad():
        logits1 = model(g).numpy()
    out = dict(src=s.astype(np.int32), dst=d.astype(np.int32), w=ww,
               meta=np.array([n, f0, hid, 9, 3, seed] + list(extra_meta), dtype=np.int64),
               logits=logits.detach().numpy(), loss=np.float32(loss.item()), logits_after_step=logits1,
               x_sum=np.float64(x.astype(np.float64).sum()), y_sum=np.int64(y.sum()))
    srng = np.random.default_rng(99)
    for i, h in enumerate(hidden):
        out[f"hidden_rows16.{i}"] = h[::16]
    for k, v in state0.items():
        out["state0_sum." + k] = np.float64(v.astype(np.float64).sum())
    for tag, dic in (("grad", grads), ("state1", state1)):
        for k, v in dic.items():
            if v.size <= 8192:
                out[f"{tag}.{k}"] = v
            else:
                idx = srng.choice(v.size, 8192, replace=False).astype(np.int64)
                out[f"{tag}_idx.{k}"] = idx
                out[f"{tag}_val.{k}"] = v.reshape(-1)[idx]
                out[f"{tag}_sum.{k}"] = np.float64(v.astype(np.float64).sum())
                out[f"{tag}_absmax.{k}"] = np.float32(np.abs(v).max())
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={n} E={len(s)} F0={f0} H={hid} loss={loss.item():.6f} -> {os.path.getsize(path) / 1e3:.0f} kB")


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    ref = _load_reference_models()
    if len(sys.argv) > 1 and sys.argv[1] == "headline":      # only the trimmed headline-shape case
        run_headline_case(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "shapes":        # only the reference's run shapes (trimmed); names: only those
        for c in SHAPE_CASES:
            if len(sys.argv) == 2 or c[0] in sys.argv[2:]:
                run_shape_case(ref, *c)
        return
    for c in SHAPE_CASES:
        run_shape_case(ref, *c)
    run_headline_case(ref)
    rng = np.random.default_rng(42)

    # (1) hand-checkable: 6 nodes, 10 edges incl. a duplicate edge, a self loop,
    #     and node 5 with in-degree 0
    src = np.array([0, 1, 2, 3, 4, 0, 0, 2, 2, 5])
    dst = np.array([1, 2, 3, 4, 0, 2, 2, 2, 0, 3])
    w = np.array([1.0, 0.5, 0.25, 0.75, 0.0, 0.5, 0.5, 1.0, 0.3, 0.9], dtype=np.float32)
    run_gcnsage_case(ref, "tiny_6n_10e", src, dst, w, 6, 3, 4, 9, 2, seed=1)

    # (2) BASELINE cfg1: one 200-word page, k-NN, F0=13, 2 layers, H=256
    s, d, ww = knn_like_graph(rng, 200, 5)
    run_gcnsage_case(ref, "page200_f13_l2", s, d, ww, 200, 13, 256, 9, 2, seed=2, bbox_like=True)

    # (3) 3 layers, class weights as in model_train.py:115-117 ('default' method)
    s, d, ww = knn_like_graph(rng, 200, 5)
    run_gcnsage_case(ref, "page200_f13_l3_cw", s, d, ww, 200, 13, 128, 9, 3, seed=3,
                     class_weights=[1, 1, 1, 1, 1, 1, 2, 1, 1])

    # (4) wide features F0=831 (BBOX+REPR+SCIBERT), smaller hidden to keep the file small
    s, d, ww = knn_like_graph(rng, 300, 5)
    run_gcnsage_case(ref, "page300_f831_l3", s, d, ww, 300, 831, 48, 9, 3, seed=4)

    # (5) directed k-NN only (bidirectional=False) -> some zero in-degree nodes possible,
    #     random multigraph with duplicates + self loops
    s, d, ww = random_graph(rng, 300, 1500)
    run_gcnsage_case(ref, "random300_multi", s, d, ww, 300, 63, 96, 9, 3, seed=5)

    # (6) unit edge weights (--edge_features=False contract: missing weights == 1.0)
    s, d, _ = knn_like_graph(rng, 150, 5, bidirect=False)
    run_gcnsage_case(ref, "page150_unitw", s, d, np.ones(len(s), np.float32), 150, 50, 32, 9, 3, seed=6)

    # (7) single node, no edges
    run_gcnsage_case(ref, "single_node", np.zeros(0, np.int64), np.zeros(0, np.int64),
                     np.zeros(0, np.float32), 1, 13, 16, 9, 2, seed=7)

    # (8) batch of heterogeneous pages (block-diagonal union, as dgl.batch)
    srcs, dsts, ws, off = [], [], [], 0
    for n in (17, 230, 64, 5, 121):
        s, d, ww = knn_like_graph(rng, n, 5)
        srcs.append(s + off), dsts.append(d + off), ws.append(ww)
        off += n
    run_gcnsage_case(ref, "batch5_hetero", np.concatenate(srcs), np.concatenate(dsts),
                     np.concatenate(ws), off, 13, 128, 9, 3, seed=8, bbox_like=True)

    # (9) MeanSAGE (fn.mean reducer), n_layers=2 -> 3 WeightedMeanSAGELayers
    s, d, ww = knn_like_graph(rng, 120, 5)
    run_meansage_case(ref, "meansage_120", s, d, ww, 120, 20, 32, 9, 2, seed=9)


if __name__ == "__main__":
    main()
