"""GAT (BASELINE cfg3, SURVEY A13 -- no reference counterpart, parity unpinned): the HIP layer against the build's own
CPU oracle (oracle/gat_cpu.py); the oracle against a dense fp64 formulation."""
import numpy as np
import pytest
import torch

from oracle import gat_cpu


def random_graph(n, e, seed):
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    dst[dst == 3] = 4                      # node 3 has no in-edge
    return src, dst


def test_oracle_gat_layer_matches_dense_fp64():
    n, f, H, D = 40, 7, 3, 5
    src, dst = random_graph(n, 300, 0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, f, generator=g, dtype=torch.float64)
    w, al, ar = (torch.randn(H * D, f, generator=g, dtype=torch.float64), torch.randn(H, D, generator=g, dtype=torch.float64),
                 torch.randn(H, D, generator=g, dtype=torch.float64))
    b = torch.randn(H * D, generator=g, dtype=torch.float64)
    got = gat_cpu.gat_layer(x, w, al, ar, b, torch.from_numpy(src), torch.from_numpy(dst), n, H)
    z = (x @ w.t()).view(n, H, D)
    want = torch.zeros(n, H, D, dtype=torch.float64)
    for v in range(n):
        es = np.nonzero(dst == v)[0]
        if len(es) == 0:
            continue
        for h in range(H):
            s = torch.stack([torch.nn.functional.leaky_relu((z[src[e], h] * al[h]).sum() + (z[v, h] * ar[h]).sum(), 0.2) for e in es])
            a = torch.softmax(s, 0)
            want[v, h] = sum(a[i] * z[src[e], h] for i, e in enumerate(es))     # duplicate edges count twice
    np.testing.assert_allclose(got.numpy(), (want.reshape(n, -1) + b).numpy(), rtol=1e-10, atol=1e-10)
    assert torch.equal(got[3], b)                                               # no in-edges: bias only


@pytest.mark.gpu
@pytest.mark.parametrize("n,e,f,hid,heads,bf16,mfma", [(300, 2500, 13, 64, 4, False, False), (257, 1500, 40, 16, 2, False, False),
                                                       (300, 2500, 13, 64, 4, True, False), (120, 900, 9, 7, 3, False, False),
                                                       (300, 2500, 13, 64, 4, True, True), (1000, 9000, 64, 64, 4, True, True),
                                                       (257, 1500, 40, 16, 2, False, True)])
def test_gat_matches_oracle_forward_and_backward(n, e, f, hid, heads, bf16, mfma):
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G
    dev = "cuda:0"
    src, dst = random_graph(n, e, n)
    torch.manual_seed(1)
    model = gte.GAT(f, hid, 9, n_layers=3, heads=heads, gather_dtype=torch.bfloat16 if bf16 else torch.float32,
                    compute_dtype=torch.bfloat16 if mfma else torch.float32)
    x = torch.randn(n, f)
    up = torch.randn(n, 9)
    params = [dict(w=l.fc.detach().clone().requires_grad_(True), a_l=l.attn_l.detach().clone().requires_grad_(True),
                   a_r=l.attn_r.detach().clone().requires_grad_(True), bias=l.bias.detach().clone().requires_grad_(True))
              for l in model.layers]
    # the oracle in fp64 at the DEVICE's rounding points (bf16 projection operands / bf16 gathered features): what is left
    # between the two is the summation order -- logits at 2e-3, every gradient at 1e-2 of its tensor's largest entry
    p64 = [{k: v.detach().double().requires_grad_(True) for k, v in p.items()} for p in params]
    want = gat_cpu.gat_forward(p64, torch.from_numpy(src), torch.from_numpy(dst), n, x.double(), heads, round_gather=bf16,
                               round_proj=mfma)
    (want * up.double()).sum().backward()
    model = model.to(dev)
    g = G.PageGraph(src, dst, n, device=dev)
    got = model(g, x.to(dev))
    (got * up.to(dev)).sum().backward()
    scale = float(want.detach().abs().max())
    tol = 2e-3 if (bf16 or mfma) else 2e-4
    assert float((got.detach().cpu().double() - want.detach()).abs().max()) <= tol * scale
    gtol = 1e-2 if (bf16 or mfma) else 1e-3
    for layer, p in zip(model.layers, p64):
        for mine, ref in ((layer.fc, p["w"]), (layer.attn_l, p["a_l"]), (layer.attn_r, p["a_r"]), (layer.bias, p["bias"])):
            r = ref.grad.numpy()
            err = float(np.abs(mine.grad.cpu().double().numpy() - r).max())
            assert err <= gtol * (np.abs(r).max() + 1e-6), (err, float(np.abs(r).max()))
    # and the bf16 configuration stays near the unrounded fp32 maths: 3 % of the largest logit through three stacked layers
    if bf16 or mfma:
        plain = gat_cpu.gat_forward([{k: v.detach() for k, v in p.items()} for p in params], torch.from_numpy(src),
                                    torch.from_numpy(dst), n, x, heads)
        assert float((got.detach().cpu() - plain).abs().max()) <= 3e-2 * float(plain.abs().max())


@pytest.mark.gpu
def test_gat_cfg3_shape_trains():
    """cfg3 shape: 4 heads x 64 hidden, 3 layers, bf16 gathers, on synthetic table-cell grid graphs."""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G, ops
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    srcs, dsts, off = [], [], 0
    for _ in range(20):                                # R x C grids, 4-neighbourhood
        R, C = int(rng.integers(3, 41)), int(rng.integers(2, 13))
        idx = np.arange(R * C).reshape(R, C)
        pairs = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]), np.stack([idx[:-1].ravel(), idx[1:].ravel()])], 1)
        pairs = np.concatenate([pairs, pairs[::-1]], 1)
        srcs.append(pairs[0] + off); dsts.append(pairs[1] + off); off += R * C
    src, dst = np.concatenate(srcs), np.concatenate(dsts)
    n = off
    torch.manual_seed(0)
    model = gte.GAT(16, 64, 5, n_layers=3, heads=4, gather_dtype=torch.bfloat16, compute_dtype=torch.bfloat16).to(dev)
    g = G.PageGraph(src, dst, n, device=dev)
    x = torch.randn(n, 16, device=dev)
    y = torch.from_numpy(rng.integers(0, 5, n)).to(dev)
    x[torch.arange(n), y] += 3.0                       # learnable
    opt = torch.optim.Adam(model.parameters(), lr=0.02)
    losses = []
    for _ in range(60):
        loss, _ = ops.cross_entropy(model(g, x), y)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses[::10]


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(1000, 256, 64), (777, 36, 256), (129, 130, 8), (5000, 256, 256), (64, 256, 13)])
def test_bf16_mfma_projection_is_exact_on_bf16_inputs(m, n, k):
    """gte_gemm_bf16_nt: with operands that are exactly representable in bf16 (small integers) fp32 accumulation is exact --
    the result must equal the fp64 product bit for bit; random operands against fp64 of the ROUNDED operands at 1e-6 relative."""
    from gnn_tableextraction_amd.components.graphs.gat import _Linear, _bf16_copy
    dev = "cuda:0"
    rng = np.random.default_rng(m + n + k)
    a = torch.from_numpy(rng.integers(-8, 9, (m, k)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.integers(-8, 9, (n, k)).astype(np.float32)).to(dev)
    got = _Linear.apply(a, w, None, True).cpu().numpy()
    np.testing.assert_array_equal(got, (a.cpu().double() @ w.cpu().double().t()).float().numpy())
    a, w = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    got = _Linear.apply(a, w, None, True).cpu().double()
    ab, wb = _bf16_copy(a)[:, :k].float().cpu().double(), _bf16_copy(w)[:, :k].float().cpu().double()
    np.testing.assert_allclose(got.numpy(), (ab @ wb.t()).numpy(), rtol=1e-5, atol=1e-5 * np.sqrt(k))
    assert torch.equal(_bf16_copy(a, 1)[:, :k].float(), torch.nn.functional.elu(a).to(torch.bfloat16).float())   # ELU + cast


@pytest.mark.gpu
@pytest.mark.parametrize("n,e,hid,heads,bf16", [(100, 4000, 64, 4, False), (100, 4000, 64, 4, True), (150, 3000, 20, 4, False),
                                                (400, 2000, 32, 3, True), (90, 2500, 5, 2, False)])
def test_gat_row_layout_kernels_match_the_lane_per_feature_kernels(n, e, hid, heads, bf16, monkeypatch):
    """csrc/gat_rows.h (head per DPP row; chosen when heads <= 4 and dim <= 64) against the lane-per-feature kernels of gat.hip
    (GTE_GAT_ROWS=0): forward output and every gradient, on graphs whose rows span several 16-edge chunks (mean in-degree
    up to 40) and rows without in-edges; same edge order, different rescaling points of the online softmax: agreement to
    2e-4 of each tensor's largest entry (three stacked layers, fp32 accumulation)."""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import graph as G
    dev = "cuda:0"
    src, dst = random_graph(n, e, n + e)
    g = G.PageGraph(src, dst, n, device=dev)
    x = torch.randn(n, 24, generator=torch.Generator().manual_seed(2)).to(dev)
    up = torch.randn(n, 9, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for rows in ("1", "0"):
        monkeypatch.setenv("GTE_GAT_ROWS", rows)
        torch.manual_seed(1)
        model = gte.GAT(24, hid, 9, n_layers=3, heads=heads, gather_dtype=torch.bfloat16 if bf16 else torch.float32).to(dev)
        out = model(g, x)
        (out * up).sum().backward()
        res[rows] = [out.detach().cpu()] + [p.grad.detach().cpu() for p in model.parameters()]
    for a, b in zip(res["1"], res["0"]):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-4 * scale, (float((a - b).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,out", [(300, 13, 256), (300, 256, 36), (1000, 64, 256), (257, 40, 32), (5000, 256, 256), (129, 8, 130)])
def test_bf16_projection_gemm_is_exact_on_bf16_operands(n, k, out):
    """gte_gemm_bf16_nt against fp32 matmul of the SAME bf16-rounded operands (products exact in fp32, so only the summation
    order differs) incl. ragged M / N / K and the branch-free store epilogue: nothing outside [n, out] is written."""
    import gnn_tableextraction_amd as gte
    from gnn_tableextraction_amd import _lib
    from gnn_tableextraction_amd.components.graphs import gat
    lib, P = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(n + k)
    x, w = torch.randn(n, k, generator=g).cuda(), torch.randn(out, k, generator=g).cuda()
    xb, wb = gat._bf16_copy(x), gat._bf16_copy(w)
    buf = torch.full((n + 3, out + 5), 7.0, device="cuda")
    z = buf[:n, :out]
    _lib.check(lib.gte_gemm_bf16_nt(P(xb), xb.stride(0), P(wb), wb.stride(0), P(z), buf.stride(0), n, out, wb.shape[1],
                                    _lib.current_stream()), "gte_gemm_bf16_nt")
    ref = xb.float() @ wb.float().t()
    assert float((z - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-5
    assert bool((buf[n:] == 7.0).all()) and bool((buf[:, out:] == 7.0).all())
