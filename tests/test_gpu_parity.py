"""GPU parity: every HIP entry point (through the C ABI) against the CPU oracle and the golden
vectors generated from the reference's models.py.  Run with ``-m gpu`` on an MI355X.

Tolerances: forward logits 1e-5 abs/rel in fp32 (BASELINE.json north_star); index/integer outputs
bit-exact; bf16 aggregation 2e-2 (storage rounding), fp32 accumulate.
"""
import glob
import os

import numpy as np
import pytest
import torch

import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G, ops
from gnn_tableextraction_amd.data import synthetic as S
from oracle import gcnsage_cpu as oc
from tests import poststep
from tests.conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["split_bf16", "f32"])
def both_gemm_modes(request):
    """The model-level parity tests run in BOTH arithmetic modes of the transform GEMMs: the default (three exact bf16 pieces per
    operand, six bf16 MFMA products -- planes GEMMs on P3 images where a layer takes them) and the fp32 MFMA mode."""
    from gnn_tableextraction_amd import ops as _ops
    prev = _ops.set_gemm_mode(request.param)
    try:
        yield request.param
    finally:
        _ops.set_gemm_mode(prev)


DEV = "cuda:0"
GCN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                   if not os.path.basename(p).startswith(("meansage", "aux_", "headline", "shape_")))
SHAPE_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "shape_*.npz")))


def dev(a, dtype=None):
    t = torch.as_tensor(a)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def test_device_is_gfx950():
    import ctypes
    lib = gte._lib.load()
    cu, wave, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    name = ctypes.create_string_buffer(64)
    assert lib.gte_device_info(ctypes.byref(cu), ctypes.byref(wave), ctypes.byref(lds), name, 64) == 0
    assert wave.value == 64 and b"gfx950" in name.value and cu.value >= 200


# ---------------------------------------------------------------- graph preparation (bit-exact)
@pytest.mark.parametrize("n,e", [(1, 0), (7, 1), (50, 400), (5000, 60000), (100000, 1200000)])
def test_coo_to_csr_bit_exact(n, e):
    rng = np.random.default_rng(n + e)
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    w = rng.random(e).astype(np.float32)
    ip, ix, wo, perm = oc.coo_to_in_csr(src, dst, n, w)
    if e == 0:
        g = G.PageGraph(src, dst, n, device=DEV)
        assert g.in_csr().indptr.cpu().tolist() == [0, 0]
        return
    indptr, indices, p, wout = ops.coo_to_csr(dev(dst, torch.int32), dev(src, torch.int32), n, dev(w))
    np.testing.assert_array_equal(indptr.cpu().numpy(), ip)
    np.testing.assert_array_equal(indices.cpu().numpy(), ix)
    np.testing.assert_array_equal(p.cpu().numpy(), perm)
    np.testing.assert_array_equal(wout.cpu().numpy(), wo)
    inv = ops.inv_degree(indptr).cpu().numpy()
    np.testing.assert_array_equal(inv, oc.in_degree_norm(ip)[:, 0])


# ---------------------------------------------------------------- aggregation
@pytest.mark.parametrize("f", [1, 3, 9, 13, 16, 63, 64, 100, 256, 512, 831, 1100])
@pytest.mark.parametrize("mean", [False, True])
def test_spmm_f32_matches_oracle(f, mean):
    rng = np.random.default_rng(f)
    n, e = 700, 5000
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    dst[dst == 5] = 6                                    # node 5: in-degree 0
    w = rng.random(e).astype(np.float32)
    x = rng.standard_normal((n, f)).astype(np.float32)
    g = oc.OracleGraph(src, dst, n, w)
    want = oc.spmm_csr_numpy(g.indptr, g.indices, g.weight, x, mean=False)
    if mean:
        want = want * g.norm
    got = ops.spmm_csr(dev(g.indptr), dev(g.indices), dev(g.weight), dev(x), n, mean=mean).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
    assert np.all(got[5] == 0)
    # unit weights (NULL pointer) == copy_u
    want1 = oc.spmm_csr_numpy(g.indptr, g.indices, None, x)
    got1 = ops.spmm_csr(dev(g.indptr), dev(g.indices), None, dev(x), n).cpu().numpy()
    np.testing.assert_allclose(got1, want1, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("f,relu", [(256, True), (128, False), (64, True), (20, True), (8, False)])
def test_spmm_accumulate_ln_epilogue_matches_the_two_kernels(f, relu):
    """gte_spmm_csr_accumulate_ln == gte_spmm_csr_accumulate followed by gte_ln_relu_fwd (z bitwise: same aggregation;
    y / stats to 1e-6: the row sums are taken in a different order)."""
    lib = gte._lib.load()
    P, cs, check = gte._lib.ptr, gte._lib.current_stream, gte._lib.check
    assert lib.gte_spmm_csr_accumulate_ln_supported(f) and lib.gte_spmm_csr_accumulate_ln_supported(258) and not lib.gte_spmm_csr_accumulate_ln_supported(1028)
    rng = np.random.default_rng(f)
    n, e = 1003, 7000
    g = oc.OracleGraph(rng.integers(0, n, e), rng.integers(0, n, e), n, rng.random(e).astype(np.float32))
    ip, ix, w = dev(g.indptr), dev(g.indices), dev(g.weight)
    t0 = dev(rng.standard_normal((n, 2 * f + 1)).astype(np.float32))          # [t_self | t_neigh | pad], ld = 2f + 1
    gam, bet = dev(1 + 0.1 * rng.standard_normal(f).astype(np.float32)), dev(0.1 * rng.standard_normal(f).astype(np.float32))
    ta, tb = t0.clone(), t0.clone()
    ya, yb = torch.empty(n, f, device=DEV), torch.empty(n, f, device=DEV)
    sa, sb = torch.empty(2 * n, device=DEV), torch.empty(2 * n, device=DEV)
    ld = ta.stride(0)
    check(lib.gte_spmm_csr_accumulate(P(ip), P(ix), P(w), P(ta) + 4 * f, ld, P(ta), ld, n, f, gte._lib.GTE_F32, 1, cs()), "acc")
    check(lib.gte_ln_relu_fwd(P(ta), ld, P(gam), P(bet), 1e-5, int(relu), P(ya), f, P(sa), n, f, cs()), "ln")
    check(lib.gte_spmm_csr_accumulate_ln(P(ip), P(ix), P(w), P(tb) + 4 * f, ld, P(tb), ld, n, f, 1, P(gam), P(bet), 1e-5,
                                         int(relu), P(yb), f, P(sb), cs()), "acc_ln")
    assert torch.equal(ta, tb)
    np.testing.assert_allclose(yb.cpu().numpy(), ya.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(sb.cpu().numpy(), sa.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_spmm_strided_rows_and_accumulate():
    rng = np.random.default_rng(1)
    n, e, f = 300, 2000, 37
    g = oc.OracleGraph(rng.integers(0, n, e), rng.integers(0, n, e), n, rng.random(e).astype(np.float32))
    big = dev(rng.standard_normal((n, 64)).astype(np.float32))
    x = big[:, 3:3 + f]                                   # ld = 64, base not 16-byte aligned
    want = oc.spmm_csr_numpy(g.indptr, g.indices, g.weight, x.cpu().numpy())
    out = torch.full((n, 50), 7.0, device=DEV)
    ops.spmm_csr(dev(g.indptr), dev(g.indices), dev(g.weight), x, n, out=out[:, 5:5 + f], accumulate=True)
    np.testing.assert_allclose(out[:, 5:5 + f].cpu().numpy(), want + 7.0, rtol=2e-6, atol=2e-6)
    assert torch.all(out[:, :5] == 7.0) and torch.all(out[:, 5 + f:] == 7.0)   # nothing outside the view


def test_spmm_is_bit_reproducible_and_degree_heavy_rows():
    rng = np.random.default_rng(2)
    n, f = 400, 256
    dst = np.concatenate([np.zeros(3000, np.int64), rng.integers(0, n, 2000)])     # row 0: 3000 in-edges
    src = rng.integers(0, n, len(dst))
    w = rng.random(len(dst)).astype(np.float32)
    x = rng.standard_normal((n, f)).astype(np.float32)
    g = oc.OracleGraph(src, dst, n, w)
    a = ops.spmm_csr(dev(g.indptr), dev(g.indices), dev(g.weight), dev(x), n)
    b = ops.spmm_csr(dev(g.indptr), dev(g.indices), dev(g.weight), dev(x), n)
    assert torch.equal(a, b)
    want = oc.spmm_csr_numpy(g.indptr, g.indices, g.weight, x.astype(np.float64))
    np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("f", [8, 13, 64, 256, 520])
def test_spmm_bf16(f):
    rng = np.random.default_rng(f)
    n, e = 500, 4000
    g = oc.OracleGraph(rng.integers(0, n, e), rng.integers(0, n, e), n, rng.random(e).astype(np.float32))
    xb = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(torch.bfloat16)
    want = oc.spmm_csr_numpy(g.indptr, g.indices, g.weight, xb.float().numpy(), mean=False) * g.norm
    got = ops.spmm_csr(dev(g.indptr), dev(g.indices), dev(g.weight), xb.to(DEV), n, mean=True)
    assert got.dtype == torch.bfloat16
    np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=2e-2, atol=2e-2)


def test_aggregate_autograd_matches_oracle_backward():
    rng = np.random.default_rng(3)
    n, e, f = 200, 1500, 48
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    w = rng.random(e).astype(np.float32)
    x = rng.standard_normal((n, f)).astype(np.float32)
    up = rng.standard_normal((n, f)).astype(np.float32)
    og = oc.OracleGraph(src, dst, n, w)
    xt = torch.from_numpy(x).requires_grad_(True)
    (oc._SpMM.apply(xt, og) * torch.from_numpy(og.norm) * torch.from_numpy(up)).sum().backward()
    g = G.PageGraph(src, dst, n, device=DEV)
    xd = dev(x).requires_grad_(True)
    (ops.aggregate(g, xd, dev(w), mean=True) * dev(up)).sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), rtol=1e-5, atol=1e-5)
    # DGL surface: update_all
    g.ndata["h"], g.edata["feat"] = dev(x), dev(w)
    g.update_all(gte.function.u_mul_e("h", "feat", "m"), gte.function.sum("m", "h"))
    want = oc.spmm_csr_numpy(og.indptr, og.indices, og.weight, x)
    np.testing.assert_allclose(g.ndata["h"].cpu().numpy(), want, rtol=2e-6, atol=2e-6)


# ---------------------------------------------------------------- MFMA GEMMs
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (33, 9, 70), (130, 256, 1662), (257, 100, 31), (9, 512, 5000),
                                   (256, 831, 20000), (300, 218, 436)])
def test_gemm_f32_all_layouts(ta, tb, m, n, k):
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
    b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    got = ops.gemm(dev(a), dev(b), trans_a=ta, trans_b=tb).cpu().numpy()
    tol = 2e-6 * np.sqrt(k) * 4 + 1e-6
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=tol * np.abs(want).max() / 4 + 1e-5)
    c0 = rng.standard_normal((m, n)).astype(np.float32)
    acc = ops.gemm(dev(a), dev(b), trans_a=ta, trans_b=tb, out=dev(c0).clone(), accumulate=True).cpu().numpy()
    np.testing.assert_allclose(acc, want + c0, rtol=1e-5, atol=tol * np.abs(want).max() / 4 + 1e-5)


def test_gemm_exact_small_integers_and_strided_views():
    # integer-valued operands: fp32 MFMA must be exact -> catches any fragment/layout mix-up
    rng = np.random.default_rng(5)
    m, n, k = 70, 45, 99
    a = rng.integers(-4, 5, (m, 128)).astype(np.float32)
    b = rng.integers(-4, 5, (k, 64)).astype(np.float32)          # asymmetric on purpose
    A = dev(a)[:, 7:7 + k]
    B = dev(b)[:, 3:3 + n]
    got = ops.gemm(A, B).cpu().numpy()
    np.testing.assert_array_equal(got, a[:, 7:7 + k] @ b[:, 3:3 + n])
    out = torch.zeros(m, 80, device=DEV)
    ops.gemm(A, B, out=out[:, 10:10 + n])
    np.testing.assert_array_equal(out[:, 10:10 + n].cpu().numpy(), got)
    assert torch.all(out[:, :10] == 0) and torch.all(out[:, 10 + n:] == 0)


@pytest.mark.parametrize("m,k1,k2,n_out,ln,relu", [
    (500, 13, 13, 256, True, True), (500, 831, 831, 256, True, True), (333, 256, 256, 9, False, False),
    (100, 64, 0, 32, True, False), (77, 20, 20, 218, True, True), (64, 50, 50, 1000, True, True),
    (40, 16, 16, 24, False, True), (3000, 96, 96, 96, True, True),
    # short K (k1 + k2 <= 64, n_out % 4 == 0, n_out <= 256): the one-pass linear + LayerNorm + ReLU kernel
    (24495, 13, 13, 256, True, True), (1001, 32, 32, 128, True, False), (7, 5, 0, 8, True, True), (2, 1, 1, 4, False, False),
    (333, 13, 13, 200, True, True), (64, 33, 32, 256, True, True)])
def test_sage_linear_fwd_vs_torch(m, k1, k2, n_out, ln, relu):
    rng = np.random.default_rng(m + n_out)
    a1 = rng.standard_normal((m, k1)).astype(np.float32)
    a2 = rng.standard_normal((m, k2)).astype(np.float32) if k2 else None
    w = (rng.standard_normal((n_out, k1 + k2)) / np.sqrt(k1 + k2)).astype(np.float32)
    b = rng.standard_normal(n_out).astype(np.float32)
    gam = (1 + 0.1 * rng.standard_normal(n_out)).astype(np.float32) if ln else None
    bet = (0.1 * rng.standard_normal(n_out)).astype(np.float32) if ln else None
    cat = torch.from_numpy(a1 if a2 is None else np.concatenate([a1, a2], 1)).double()
    z = torch.nn.functional.linear(cat, torch.from_numpy(w).double(), torch.from_numpy(b).double())
    want = z
    if ln:
        want = torch.nn.functional.layer_norm(z, (n_out,), torch.from_numpy(gam).double(), torch.from_numpy(bet).double(), 1e-5)
    if relu:
        want = torch.relu(want)
    y, zs, st = ops.sage_linear_fwd(dev(a1), None if a2 is None else dev(a2), dev(w), dev(b),
                                    None if gam is None else dev(gam), None if bet is None else dev(bet),
                                    1e-5, relu, save_for_backward=True)
    np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    if ln:
        np.testing.assert_allclose(zs.cpu().numpy(), z.numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(st[:m].cpu().numpy(), z.mean(1).numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(st[m:2 * m].cpu().numpy(), (z.var(1, unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-4)


@pytest.mark.parametrize("ta,tb,m,n,k", [(False, True, 128 * 264, 128, 1024), (False, False, 64 * 520 - 5, 128, 831),
                                          (False, True, 128 * 259 - 77, 256, 300)])
def test_gemm_tail_split_matches_unsplit_and_float64(ta, tb, m, n, k):
    """GEMM tail split (gte_gemm_set_tail_workspace): launches whose last round holds few tiles cut those tiles' K range
    into pieces + a fix-up launch.  Same result as the unsplit launch up to summation order, and against float64."""
    lib = gte._lib.load()
    rng = np.random.default_rng(m + n + k)
    a = dev(rng.standard_normal((m, k)).astype(np.float32))
    b = dev((rng.standard_normal((n, k) if tb else (k, n)) / np.sqrt(k)).astype(np.float32))
    plain = ops.gemm(a, b, trans_b=tb)
    ws = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=DEV)
    gte._lib.check(lib.gte_gemm_set_tail_workspace(gte._lib.ptr(ws), ws.numel()), "set")
    try:
        split = ops.gemm(a, b, trans_b=tb)
        again = ops.gemm(a, b, trans_b=tb)
    finally:
        lib.gte_gemm_set_tail_workspace(None, 0)
    assert torch.equal(split, again)                                   # deterministic
    assert not torch.equal(split, plain) or m * n < 1                  # the split path really ran (different summation order)
    want = a.double().cpu() @ (b.double().cpu().T if tb else b.double().cpu())
    np.testing.assert_allclose(split.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(plain.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


def test_gemm_tail_split_keeps_the_epilogue_semantics():
    """bias (whole / first N segment only) and ReLU are applied by the fix-up launch exactly as by the GEMM epilogue."""
    lib = gte._lib.load()
    P, cs, check = gte._lib.ptr, gte._lib.current_stream, gte._lib.check
    rng = np.random.default_rng(7)
    m, k, n_out = 128 * 264 - 3, 256, 128                    # 264 row tiles: the last 8 tiles split over the idle CUs
    a1, a2 = dev(rng.standard_normal((m, k)).astype(np.float32)), dev(rng.standard_normal((m, k)).astype(np.float32))
    w = dev((rng.standard_normal((n_out, 2 * k)) / np.sqrt(2 * k)).astype(np.float32))
    b = dev(rng.standard_normal(n_out).astype(np.float32))
    ws = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=DEV)

    def run(split):
        check(lib.gte_gemm_set_tail_workspace(P(ws) if split else None, ws.numel() if split else 0), "set")
        try:
            y, _, _ = ops.sage_linear_fwd(a1, a2, w, b, None, None, 1e-5, True, False)          # bias + ReLU, no LayerNorm
            t = torch.empty(m, 2 * n_out, device=DEV)
            check(lib.gte_sage_transform_fwd(P(a1), k, k, P(w), 2 * k, P(b), n_out, P(t), 2 * n_out, m, cs()), "transform")
        finally:
            lib.gte_gemm_set_tail_workspace(None, 0)
        return y, t
    y0, t0 = run(False)
    y1, t1 = run(True)
    want_y = torch.relu(torch.cat([a1, a2], 1).double().cpu() @ w.double().cpu().T + b.double().cpu())
    want_t = torch.cat([a1.double().cpu() @ w.double().cpu()[:, :k].T + b.double().cpu(), a1.double().cpu() @ w.double().cpu()[:, k:].T], 1)
    for got in (y0, y1):
        np.testing.assert_allclose(got.cpu().numpy(), want_y.numpy(), rtol=1e-5, atol=1e-5)
    for got in (t0, t1):
        np.testing.assert_allclose(got.cpu().numpy(), want_t.numpy(), rtol=1e-5, atol=1e-5)
    assert not torch.equal(y0, y1)                           # the split path ran (different summation order in the tail tiles)


@pytest.mark.parametrize("n,f,o", [(1000, 831, 256), (513, 256, 256), (300, 13, 40), (77, 50, 33), (5, 3, 2), (24495, 831, 256)])
def test_qform_entry_points_vs_float64(n, f, o):
    """gte_sage_transform_fwd / gte_sage_qform_dw / gte_sage_qform_dx (transform-then-aggregate form of a layer)
    against float64 with strided operands: t = [x W_s^T + b | x W_n^T], dW = [dz^T x | q^T x], dx = dz W_s + q W_n."""
    lib = gte._lib.load()
    P, cs, check = gte._lib.ptr, gte._lib.current_stream, gte._lib.check
    rng = np.random.default_rng(n + f + o)
    xbuf = dev(rng.standard_normal((n, f + 1)).astype(np.float32))
    x = xbuf[:, :f]
    W = dev((rng.standard_normal((o, 2 * f)) / np.sqrt(2 * f)).astype(np.float32))
    b = dev(rng.standard_normal(o).astype(np.float32))
    t = torch.full((n, 2 * o + 3), 5.0, device=DEV)
    check(lib.gte_sage_transform_fwd(P(x), x.stride(0), f, P(W), 2 * f, P(b), o, P(t), t.stride(0), n, cs()), "transform_fwd")
    xd, Wd = x.double().cpu(), W.double().cpu()
    want_t = torch.cat([xd @ Wd[:, :f].T + b.double().cpu(), xd @ Wd[:, f:].T], 1)
    np.testing.assert_allclose(t[:, :2 * o].cpu().numpy(), want_t.numpy(), rtol=1e-5, atol=1e-5)
    assert float(t[:, 2 * o:].min()) == 5.0
    dzq = dev(rng.standard_normal((n, 2 * o)).astype(np.float32) / n)          # dz | q side by side: ld = 2 o
    dz, q = dzq[:, :o], dzq[:, o:]
    dW = torch.full((o, 2 * f + 2), 5.0, device=DEV)
    ws = torch.empty(max(int(lib.gte_sage_qform_dw_workspace_bytes(o, f, n)), 256), dtype=torch.uint8, device=DEV)
    check(lib.gte_sage_qform_dw(P(dz), 2 * o, P(q), 2 * o, P(x), x.stride(0), f, P(dW), dW.stride(0), o, n, P(ws), ws.numel(),
                                cs()), "qform_dw")
    dzd, qd = dz.double().cpu(), q.double().cpu()
    want_dW = torch.cat([dzd.T @ xd, qd.T @ xd], 1)
    tol = lambda ref: dict(rtol=1e-5, atol=2e-6 * float(ref.abs().max()) + 1e-9)
    np.testing.assert_allclose(dW[:, :2 * f].cpu().numpy(), want_dW.numpy(), **tol(want_dW))
    assert float(dW[:, 2 * f:].min()) == 5.0
    dx = torch.full((n, f + 2), 5.0, device=DEV)
    check(lib.gte_sage_qform_dx(P(dz), 2 * o, P(q), 2 * o, P(W), 2 * f, f, o, P(dx), dx.stride(0), n, cs()), "qform_dx")
    want_dx = dzd @ Wd[:, :f] + qd @ Wd[:, f:]
    np.testing.assert_allclose(dx[:, :f].cpu().numpy(), want_dx.numpy(), **tol(want_dx))
    assert float(dx[:, f:].min()) == 5.0


@pytest.mark.parametrize("n,f,c,relu", [(1000, 256, 9, True), (333, 64, 4, False), (24495, 256, 9, True), (77, 128, 16, True),
                                        (31, 8, 1, True), (2049, 200, 12, True)])
def test_narrow_fwd_with_fused_layernorm_forward(n, f, c, relu):
    """gte_sage_narrow_fwd_ln (LayerNorm + ReLU of the layer below inside the output layer's forward) against float64 and
    against the two separate launches it replaces (gte_ln_relu_fwd + gte_sage_narrow_fwd)."""
    lib = gte._lib.load()
    P, cs, check = gte._lib.ptr, gte._lib.current_stream, gte._lib.check
    rng = np.random.default_rng(n + f + c)
    zbuf = dev((rng.standard_normal((n, f + 4)) * 1.5 + 0.3).astype(np.float32))
    z = zbuf[:, :f]                                                   # row stride > width
    gam, bet = dev((1 + 0.1 * rng.standard_normal(f)).astype(np.float32)), dev((0.1 * rng.standard_normal(f)).astype(np.float32))
    W = dev((rng.standard_normal((c, 2 * f)) / np.sqrt(2 * f)).astype(np.float32))
    bias = dev(rng.standard_normal(c).astype(np.float32))
    y = torch.full((n, f), 7.0, device=DEV); stats = torch.zeros(2 * n, device=DEV)
    ts, tn = torch.zeros(n, c, device=DEV), torch.zeros(n, c, device=DEV)
    check(lib.gte_sage_narrow_fwd_ln(P(z), z.stride(0), f, P(gam), P(bet), 1e-5, int(relu), P(y), f, P(stats), P(W), 2 * f, P(bias),
                                     c, P(ts), c, P(tn), c, n, cs()), "narrow_fwd_ln")
    zd = z.double().cpu()
    want_y = torch.nn.functional.layer_norm(zd, (f,), gam.double().cpu(), bet.double().cpu(), 1e-5)
    if relu:
        want_y = torch.relu(want_y)
    Wd = W.double().cpu()
    np.testing.assert_allclose(y.cpu().numpy(), want_y.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(stats[:n].cpu().numpy(), zd.mean(1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(stats[n:].cpu().numpy(), (zd.var(1, unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-4)
    np.testing.assert_allclose(ts.cpu().numpy(), (want_y @ Wd[:, :f].T + bias.double().cpu()).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tn.cpu().numpy(), (want_y @ Wd[:, f:].T).numpy(), rtol=1e-5, atol=1e-5)
    # the two launches it replaces
    y2 = torch.empty(n, f, device=DEV); st2 = torch.zeros(2 * n, device=DEV)
    ts2, tn2 = torch.zeros(n, c, device=DEV), torch.zeros(n, c, device=DEV)
    check(lib.gte_ln_relu_fwd(P(z), z.stride(0), P(gam), P(bet), 1e-5, int(relu), P(y2), f, P(st2), n, f, cs()), "ln")
    check(lib.gte_sage_narrow_fwd(P(y2), f, f, P(W), 2 * f, P(bias), c, P(ts2), c, P(tn2), c, n, cs()), "narrow_fwd")
    np.testing.assert_allclose(y.cpu().numpy(), y2.cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ts.cpu().numpy(), ts2.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,f,c,relu", [(1000, 256, 9, True), (333, 64, 4, False), (24495, 256, 9, True), (77, 128, 16, True), (1, 256, 9, True)])
def test_narrow_bwd_with_layernorm_backward_in_row_form_is_bitwise_the_two_launches(n, f, c, relu):
    """gte_sage_narrow_bwd_ln_p3 (the dh tile of a row block through LDS, whole rows in the layout and with the arithmetic of
    gte_ln_relu_bwd): dz of the layer below as fp32 and as a P3 image bit for bit gte_sage_narrow_bwd + gte_ln_relu_bwd_p3; z is a
    strided view (the left half of t = [t_self | t_neigh]); the column sums agree to summation order."""
    lib = gte._lib.load()
    P, cs, check = gte._lib.ptr, gte._lib.current_stream, gte._lib.check
    rng = np.random.default_rng(n + f)
    new = lambda *s: torch.empty(*s, device=DEV)
    t = dev(rng.standard_normal((n, 2 * f)).astype(np.float32))
    z = t[:, :f]
    gam, bet = dev(1 + 0.1 * rng.standard_normal(f).astype(np.float32)), dev(0.1 * rng.standard_normal(f).astype(np.float32))
    h, stats = new(n, f), new(2 * n)
    check(lib.gte_ln_relu_fwd(P(z), 2 * f, P(gam), P(bet), 1e-5, int(relu), P(h), f, P(stats), n, f, cs()), "ln fwd")
    W = dev((rng.standard_normal((c, 2 * f)) / np.sqrt(2 * f)).astype(np.float32))
    dl, q = dev(rng.standard_normal((n, c)).astype(np.float32) / n), dev(rng.standard_normal((n, c)).astype(np.float32) / n)
    wsn = torch.empty(int(lib.gte_sage_narrow_bwd_workspace_bytes(n, f, c)), dtype=torch.uint8, device=DEV)
    # two launches
    dh, dWa, dba = new(n, f), new(c, 2 * f), new(c)
    check(lib.gte_sage_narrow_bwd(P(dl), c, P(q), c, P(h), f, f, P(W), 2 * f, c, P(dh), f, P(dWa), 2 * f, P(dba), n, P(wsn),
                                  wsn.numel(), cs()), "bwd")
    img = f % 16 == 0 and f >= 128
    dga, dbea, dbia = new(f), new(f), new(f)
    dzp_a = ops.P3.empty(n, f, DEV) if img else None
    wl = torch.empty(int(lib.gte_ln_relu_bwd_workspace_bytes(n, f)), dtype=torch.uint8, device=DEV)
    if img:
        check(lib.gte_ln_relu_bwd_p3(P(dh), f, P(z), 2 * f, P(stats), P(gam), P(bet), int(relu), P(dh), f, P(dzp_a.data), dzp_a.ldp,
                                     P(dga), P(dbea), P(dbia), n, f, P(wl), wl.numel(), cs()), "ln bwd p3")
    else:
        check(lib.gte_ln_relu_bwd(P(dh), f, P(z), 2 * f, P(stats), P(gam), P(bet), int(relu), P(dh), f, P(dga), P(dbea), P(dbia), n, f,
                                  P(wl), wl.numel(), cs()), "ln bwd")
    # one launch
    dzb, dWb, dbb, dgb, dbeb, dbib = torch.full((n, f), 7.0, device=DEV), new(c, 2 * f), new(c), new(f), new(f), new(f)
    dzp_b = ops.P3.empty(n, f, DEV) if f % 16 == 0 else None
    wln = torch.empty(int(lib.gte_sage_narrow_bwd_ln_workspace_bytes(n, f)), dtype=torch.uint8, device=DEV)
    check(lib.gte_sage_narrow_bwd_ln_p3(P(dl), c, P(q), c, P(h), f, f, P(W), 2 * f, c, P(dzb), f,
                                        P(dzp_b.data) if dzp_b is not None else None, dzp_b.ldp if dzp_b is not None else 0,
                                        P(dWb), 2 * f, P(dbb), n, P(wsn), wsn.numel(), None, 1.0, None, P(z), 2 * f, P(stats), P(gam),
                                        P(bet), int(relu), P(dgb), P(dbeb), P(dbib), P(wln), wln.numel(), cs()),
          "bwd_ln_p3")
    if img:
        assert torch.equal(dzb, dh)
    else:           # narrow rows: the separate launch runs another LayerNorm backward kernel (same maths, other instruction order)
        np.testing.assert_allclose(dzb.cpu().numpy(), dh.cpu().numpy(), rtol=2e-5, atol=3e-6 * float(dh.abs().max()))
    if dzp_b is not None:
        assert torch.equal(ops.p3_to_f32(dzp_b), dzb)
    if img:
        assert torch.equal(dzp_b.data, dzp_a.data)
    for got, want in ((dWb, dWa), (dbb, dba), (dgb, dga), (dbeb, dbea), (dbib, dbia)):
        ref = want.cpu().numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=3e-6 * np.abs(ref).max() + 1e-12)


@pytest.mark.parametrize("n,f,c,weighted,flt", [(1500, 256, 9, True, True), (777, 64, 4, False, False), (65, 8, 16, True, False),
                                                  (24495, 256, 9, False, True)])
def test_fused_head_matches_the_separate_kernels(n, f, c, weighted, flt):
    """gte_head_agg_ce + gte_sage_narrow_bwd_ce (aggregation + CE + unnormalised gradient in one launch, 1/sum(w) and the
    loss inside the backward kernel) against gte_spmm_csr_accumulate + gte_weighted_ce + gte_sage_narrow_bwd."""
    lib = gte._lib.load()
    P, cs, check = gte._lib.ptr, gte._lib.current_stream, gte._lib.check
    assert lib.gte_head_supported(f, c) and not lib.gte_head_supported(12, 9)
    rng = np.random.default_rng(n + c)
    e = 6 * n
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    dst[dst == 3] = 4                                                    # node 3: no in-edges
    w = rng.random(e).astype(np.float32)
    g = G.PageGraph(src, dst, n, device=DEV)
    csr, rcsr = g.in_csr(), g.out_csr()
    w_in, w_out = g.in_weights(dev(w)), g.out_weights(dev(w), True)
    h = dev(rng.standard_normal((n, f)).astype(np.float32))
    W = dev((rng.standard_normal((c, 2 * f)) / np.sqrt(2 * f)).astype(np.float32))
    ts0, tn = dev(rng.standard_normal((n, c)).astype(np.float32)), dev(rng.standard_normal((n, c)).astype(np.float32))
    y = rng.integers(0, c, n)
    lab = dev(y.astype(np.float32)) if flt else dev(y.astype(np.int64))
    cw = dev(rng.random(c).astype(np.float32) + 0.5) if weighted else None
    gs = 0.37
    new = lambda *s: torch.empty(*s, device=DEV)
    # separate kernels
    la = ts0.clone()
    check(lib.gte_spmm_csr_accumulate(P(csr.indptr), P(csr.indices), P(w_in), P(tn), c, P(la), c, n, c, gte._lib.GTE_F32, 1, cs()), "agg")
    dla, outa = new(n, c), new(3)
    wsa = torch.empty(int(lib.gte_weighted_ce_workspace_bytes(n)), dtype=torch.uint8, device=DEV)
    check(lib.gte_weighted_ce(P(la), c, P(lab), int(flt), P(cw), n, c, gs, P(dla), c, P(outa), P(wsa), wsa.numel(), cs()), "ce")
    qa = ops.spmm_csr(rcsr.indptr, rcsr.indices, w_out, dla, n)
    wsn = torch.empty(int(lib.gte_sage_narrow_bwd_workspace_bytes(n, f, c)), dtype=torch.uint8, device=DEV)
    dha, dWa, dba = new(n, f), new(c, 2 * f), new(c)
    check(lib.gte_sage_narrow_bwd(P(dla), c, P(qa), c, P(h), f, f, P(W), 2 * f, c, P(dha), f, P(dWa), 2 * f, P(dba), n, P(wsn),
                                  wsn.numel(), cs()), "bwd")
    # fused head
    lb, dlb, outb = ts0.clone(), new(n, c), new(3)
    part = torch.empty(int(lib.gte_head_agg_ce_workspace_bytes(n)), dtype=torch.uint8, device=DEV)
    check(lib.gte_head_agg_ce(P(csr.indptr), P(csr.indices), P(w_in), P(tn), c, P(lb), c, P(lab), int(flt), P(cw), n, c, 1, P(dlb),
                              c, P(part), part.numel(), cs()), "head")
    qb = ops.spmm_csr(rcsr.indptr, rcsr.indices, w_out, dlb, n)
    dhb, dWb, dbb = new(n, f), new(c, 2 * f), new(c)
    check(lib.gte_sage_narrow_bwd_ce(P(dlb), c, P(qb), c, P(h), f, f, P(W), 2 * f, c, P(dhb), f, P(dWb), 2 * f, P(dbb), n, P(wsn),
                                     wsn.numel(), P(part), gs, P(outb), cs()), "bwd_ce")
    assert torch.equal(la, lb)                                             # same aggregation, bit for bit
    np.testing.assert_allclose(outb.cpu().numpy(), outa.cpu().numpy(), rtol=2e-6, atol=0)
    assert float(outb[2]) == float(outa[2])                                # identical arg-max decisions
    for got, want in ((dhb, dha), (dWb, dWa), (dbb, dba)):
        ref = want.cpu().numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-6 * np.abs(ref).max() + 1e-12)


@pytest.mark.parametrize("n,f,c", [(1000, 256, 9), (777, 64, 4), (333, 200, 16), (50, 13, 9), (2049, 128, 12), (31, 8, 1),
                                   (24495, 256, 9)])
def test_narrow_layer_fwd_bwd_vs_float64(n, f, c):
    """The class-count-wide output layer (gte_sage_narrow_fwd / _bwd: matrix-pipe kernels when F % 8 == 0, plain
    FMA kernels otherwise) against float64: t_self = h W_s^T + b, t_neigh = h W_n^T; dh = dl W_s + q W_n,
    dW = [dl^T h | q^T h], dbias = colsum(dl).  Strided h / dh / dW exercise the leading dimensions."""
    lib = gte._lib.load()
    P, cs = gte._lib.ptr, gte._lib.current_stream
    assert lib.gte_sage_narrow_supported(f, c)
    rng = np.random.default_rng(n + f + c)
    hbuf = dev(rng.standard_normal((n, f + 3)).astype(np.float32))
    h = hbuf[:, :f]
    W = dev((rng.standard_normal((c, 2 * f)) / np.sqrt(2 * f)).astype(np.float32))
    b = dev(rng.standard_normal(c).astype(np.float32))
    ts, tn = torch.full((n, c), 7.0, device=DEV), torch.full((n, c), 7.0, device=DEV)
    gte._lib.check(lib.gte_sage_narrow_fwd(P(h), h.stride(0), f, P(W), W.stride(0), P(b), c, P(ts), c, P(tn), c, n, cs()), "fwd")
    hd, Wd = h.double().cpu(), W.double().cpu()
    np.testing.assert_allclose(ts.cpu().numpy(), (hd @ Wd[:, :f].T + b.double().cpu()).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tn.cpu().numpy(), (hd @ Wd[:, f:].T).numpy(), rtol=1e-5, atol=1e-5)

    dl = dev(rng.standard_normal((n, c)).astype(np.float32) / n)
    q = dev(rng.standard_normal((n, c)).astype(np.float32) / n)
    dhbuf = torch.full((n, f + 5), 3.0, device=DEV)
    dh = dhbuf[:, :f]
    dWbuf = torch.full((c, 2 * f + 2), 3.0, device=DEV)
    dW = dWbuf[:, :2 * f]
    db = torch.full((c,), 3.0, device=DEV)
    ws = torch.empty(lib.gte_sage_narrow_bwd_workspace_bytes(n, f, c), dtype=torch.uint8, device=DEV)
    gte._lib.check(lib.gte_sage_narrow_bwd(P(dl), c, P(q), c, P(h), h.stride(0), f, P(W), W.stride(0), c, P(dh), dh.stride(0),
                                           P(dW), dW.stride(0), P(db), n, P(ws), ws.numel(), cs()), "bwd")
    dld, qd = dl.double().cpu(), q.double().cpu()
    want_dh = dld @ Wd[:, :f] + qd @ Wd[:, f:]
    want_dW = torch.cat([dld.T @ hd, qd.T @ hd], 1)
    tol = lambda ref: dict(rtol=1e-5, atol=2e-6 * float(ref.abs().max()) + 1e-9)
    np.testing.assert_allclose(dh.cpu().numpy(), want_dh.numpy(), **tol(want_dh))
    np.testing.assert_allclose(dW.cpu().numpy(), want_dW.numpy(), **tol(want_dW))
    np.testing.assert_allclose(db.cpu().numpy(), dld.sum(0).numpy(), **tol(dld.sum(0)))
    assert float(dhbuf[:, f:].min()) == 3.0 and float(dWbuf[:, 2 * f:].min()) == 3.0          # padding untouched
    # dh == nullptr (first layer): dW / dbias only
    dW2, db2 = torch.empty(c, 2 * f, device=DEV), torch.empty(c, device=DEV)
    gte._lib.check(lib.gte_sage_narrow_bwd(P(dl), c, P(q), c, P(h), h.stride(0), f, P(W), W.stride(0), c, None, f,
                                           P(dW2), 2 * f, P(db2), n, P(ws), ws.numel(), cs()), "bwd")
    assert torch.equal(dW2, dW.contiguous()) and torch.equal(db2, db)


@pytest.mark.parametrize("m,n,ln,relu", [(300, 256, True, True), (100, 9, False, False), (513, 218, True, True),
                                         (64, 40, False, True), (50, 1000, True, False), (31, 64, True, True),
                                         (4099, 512, True, True), (777, 128, True, False), (2500, 256, False, True)])
def test_ln_relu_bwd_vs_torch_autograd(m, n, ln, relu):
    rng = np.random.default_rng(m + n)
    z = torch.from_numpy(rng.standard_normal((m, n))).double().requires_grad_(True)
    gam = torch.from_numpy(1 + 0.1 * rng.standard_normal(n)).double().requires_grad_(True)
    bet = torch.from_numpy(0.1 * rng.standard_normal(n)).double().requires_grad_(True)
    dy = rng.standard_normal((m, n)).astype(np.float32)
    y = torch.nn.functional.layer_norm(z, (n,), gam, bet, 1e-5) if ln else z
    if relu:
        y = torch.relu(y)
    y.backward(torch.from_numpy(dy).double())
    zd = dev(z.detach().float())
    stats = None
    if ln:
        mu = z.detach().mean(1)
        rstd = 1.0 / torch.sqrt(z.detach().var(1, unbiased=False) + 1e-5)
        stats = dev(torch.cat([mu, rstd]).float())
    dg = torch.zeros(n, device=DEV) if ln else None
    db = torch.zeros(n, device=DEV) if ln else None
    dbias = torch.zeros(n, device=DEV)
    dz = ops.ln_relu_bwd(dev(dy), zd, stats, dev(gam.detach().float()) if ln else None,
                         dev(bet.detach().float()) if ln else None, relu, dg, db, dbias)
    np.testing.assert_allclose(dz.cpu().numpy(), z.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dbias.cpu().numpy(), z.grad.sum(0).numpy(), rtol=1e-4, atol=1e-4)
    if ln:
        np.testing.assert_allclose(dg.cpu().numpy(), gam.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(db.cpu().numpy(), bet.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("m,k1,k2,n_out,relu", [(24495, 13, 13, 256, True), (1000, 13, 13, 256, False), (333, 9, 9, 128, True),
                                                (65, 14, 14, 200, True), (7, 5, 0, 8, True), (64, 1, 1, 4, True), (4099, 8, 8, 64, True)])
def test_smallk_input_layer_backward_in_one_pass(m, k1, k2, n_out, relu):
    """gte_sage_smallk_bwd (z recomputed from the 2 F0 inputs, dz never stored) against fp64 autograd of
    relu(LayerNorm([x | ahn] W^T + b)) and against the two-kernel path it replaces (gte_ln_relu_bwd + gte_sage_linear_dw on the
    z the forward saved)."""
    from gnn_tableextraction_amd import _lib
    lib, P = _lib.load(), _lib.ptr
    if os.environ.get("GTE_SMALLK_BWD", "1")[0] == "0":
        pytest.skip("the one-pass backward is switched off")
    assert lib.gte_sage_smallk_bwd_supported(k1 + k2, n_out) == 1
    rng = np.random.default_rng(m + n_out)
    a1 = rng.standard_normal((m, k1)).astype(np.float32)
    a2 = rng.standard_normal((m, k2)).astype(np.float32) if k2 else None
    w = (rng.standard_normal((n_out, k1 + k2)) / np.sqrt(k1 + k2)).astype(np.float32)
    b = rng.standard_normal(n_out).astype(np.float32)
    gam = (1 + 0.1 * rng.standard_normal(n_out)).astype(np.float32)
    bet = (0.1 * rng.standard_normal(n_out)).astype(np.float32)
    dy = rng.standard_normal((m, n_out)).astype(np.float32)
    # forward on the device (one-pass kernel; z saved only for the two-kernel comparison)
    d1, d2, dw_, db_, dg_, dbe_, ddy = dev(a1), (None if a2 is None else dev(a2)), dev(w), dev(b), dev(gam), dev(bet), dev(dy)
    z = torch.empty(m, n_out, device=DEV)
    yd = torch.empty(m, n_out, device=DEV)
    stats = torch.empty(2 * m, device=DEV)
    st = _lib.current_stream()
    _lib.check(lib.gte_sage_linear_fwd(P(d1), k1, k1, P(d2), k2, k2, P(dw_), k1 + k2, P(db_), P(dg_), P(dbe_), 1e-5, int(relu),
                                       P(z), n_out, P(stats), P(yd), n_out, m, n_out, st), "fwd")
    # ... and without saving z: same y and stats
    y2, stats2 = torch.empty_like(yd), torch.empty_like(stats)
    _lib.check(lib.gte_sage_linear_fwd(P(d1), k1, k1, P(d2), k2, k2, P(dw_), k1 + k2, P(db_), P(dg_), P(dbe_), 1e-5, int(relu),
                                       None, n_out, P(stats2), P(y2), n_out, m, n_out, st), "fwd")
    assert torch.equal(y2, yd) and torch.equal(stats2, stats)
    # fp64 autograd, the ReLU mask taken from the device's forward (of ~6 M outputs a few lie within rounding of 0: their mask
    # is the forward's to decide, and each one moves a whole row of dW by O(1))
    cat = torch.from_numpy(a1 if a2 is None else np.concatenate([a1, a2], 1)).double()
    W64, b64 = torch.from_numpy(w).double().requires_grad_(True), torch.from_numpy(b).double().requires_grad_(True)
    g64, be64 = torch.from_numpy(gam).double().requires_grad_(True), torch.from_numpy(bet).double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(torch.nn.functional.linear(cat, W64, b64), (n_out,), g64, be64, 1e-5)
    if relu:
        y = y * (yd > 0).double().cpu()
    y.backward(torch.from_numpy(dy).double())
    gW = torch.full((n_out, k1 + k2), 7.0, device=DEV)
    gb, gg, gbe = (torch.full((n_out,), 7.0, device=DEV) for _ in range(3))
    ops.sage_smallk_bwd(ddy, d1, d2, dw_, db_, dg_, dbe_, stats, relu, gW, gb, gg, gbe)
    scale = lambda t: float(t.abs().max())
    np.testing.assert_allclose(gW.cpu().numpy(), W64.grad.numpy(), rtol=1e-4, atol=1e-5 * scale(W64.grad) + 1e-6)
    np.testing.assert_allclose(gb.cpu().numpy(), b64.grad.numpy(), rtol=1e-4, atol=1e-5 * scale(b64.grad) + 1e-6)
    np.testing.assert_allclose(gg.cpu().numpy(), g64.grad.numpy(), rtol=1e-4, atol=1e-5 * scale(g64.grad) + 1e-6)
    np.testing.assert_allclose(gbe.cpu().numpy(), be64.grad.numpy(), rtol=1e-4, atol=1e-5 * scale(be64.grad) + 1e-6)
    # the two-kernel path on the saved z: same dz values (same arithmetic), sums in another order
    dg2, db2, dbias2 = (torch.zeros(n_out, device=DEV) for _ in range(3))
    dz = ops.ln_relu_bwd(ddy, z, stats, dg_, dbe_, relu, dg2, db2, dbias2)
    gW2 = ops.sage_linear_dw(dz, d1, d2, torch.empty(n_out, k1 + k2, device=DEV))
    np.testing.assert_allclose(gW.cpu().numpy(), gW2.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale(gW2) + 1e-7)
    np.testing.assert_allclose(gg.cpu().numpy(), dg2.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale(dg2) + 1e-7)
    np.testing.assert_allclose(gb.cpu().numpy(), dbias2.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale(dbias2) + 1e-7)


# ---------------------------------------------------------------- loss + optimiser
@pytest.mark.parametrize("n,c,weighted,float_labels", [(1, 9, False, False), (1000, 9, True, False),
                                                       (7777, 9, False, True), (300, 13, True, True)])
def test_weighted_ce_vs_torch(n, c, weighted, float_labels):
    rng = np.random.default_rng(n)
    logits = (3 * rng.standard_normal((n, c))).astype(np.float32)
    y = rng.integers(0, c, n)
    cw = (0.5 + rng.random(c)).astype(np.float32) if weighted else None
    lt = torch.from_numpy(logits).requires_grad_(True)
    loss = torch.nn.CrossEntropyLoss(weight=None if cw is None else torch.from_numpy(cw))(lt, torch.from_numpy(y))
    loss.backward()
    labels = dev(y.astype(np.float32)) if float_labels else dev(y)
    out3, dl = ops.weighted_ce(dev(logits), labels, None if cw is None else dev(cw))
    o = out3.cpu().numpy()
    assert abs(o[0] - loss.item()) < 1e-5 * max(1, abs(loss.item()))
    assert int(o[2]) == int((logits.argmax(1) == y).sum())
    np.testing.assert_allclose(dl.cpu().numpy(), lt.grad.numpy(), rtol=1e-5, atol=1e-7)


def test_adam_step_vs_torch_adam():
    rng = np.random.default_rng(0)
    n = 100003
    p0 = rng.standard_normal(n).astype(np.float32)
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=0.01, weight_decay=5e-4)
    p, m, v = dev(p0.copy()), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        ops.adam_step(p, dev(g), m, v, step, lr=0.01, weight_decay=5e-4)
    np.testing.assert_allclose(p.cpu().numpy(), pt.detach().numpy(), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- whole model vs the reference's golden vectors
def load_model(z):
    n, f0, hid, ncls, nl, seed = (int(v) for v in z["meta"])
    m = gte.GcnSAGE(f0, hid, ncls, nl, torch.nn.functional.relu, 0)
    m.load_state_dict({k[len("state0."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state0.")})
    g = G.PageGraph(z["src"], z["dst"], n, device=DEV)
    g.ndata["feat"] = dev(z["x"])
    g.edata["feat"] = dev(z["w"])
    return m.to(DEV), g


@pytest.mark.usefixtures("both_gemm_modes")
@pytest.mark.parametrize("name", GCN_CASES)
def test_gcnsage_forward_matches_reference_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    model, g = load_model(z)
    hidden = []
    hooks = [l.register_forward_hook(lambda m, i, o: hidden.append(o.detach().cpu().numpy())) for l in model.layers]
    with torch.no_grad():
        logits = model(g).cpu().numpy()
    for h in hooks:
        h.remove()
    np.testing.assert_allclose(logits, z["logits"], rtol=1e-5, atol=1e-5)       # north_star tolerance
    for i, h in enumerate(hidden):
        np.testing.assert_allclose(h, z[f"hidden.{i}"], rtol=1e-5, atol=2e-5)
    # tensor-level call signature: (node_feats, edge_index[, edge_weight]) -> logits
    ei = torch.stack([dev(z["src"]), dev(z["dst"])])
    with torch.no_grad():
        l2 = model(dev(z["x"]), ei, dev(z["w"])).cpu().numpy()
    np.testing.assert_array_equal(l2, logits)


@pytest.mark.usefixtures("both_gemm_modes")
@pytest.mark.parametrize("name", GCN_CASES)
def test_gcnsage_train_step_matches_reference_golden(name):
    """loss.backward() through the HIP autograd nodes + torch.optim.Adam, exactly as model_train.py:320-332
    drives the model; gradients and the post-step logits against the reference's."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    model, g = load_model(z)
    cw = dev(z["class_weights"]) if "class_weights" in z.files else None
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    logits = model(g)
    loss = torch.nn.CrossEntropyLoss(weight=cw)(logits, dev(z["y"]))
    opt.zero_grad()
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    for k, p in model.named_parameters():
        ref = z["grad." + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(ref).max())
    opt.step()
    with torch.no_grad():
        after = model(g).cpu().numpy()
    # parameters with a resolved gradient equal the reference's to 1e-5, post-step logits at 1e-4 (tests/poststep.py)
    poststep.check_against_fixture(z, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, after)
    # same step with the HIP loss: identical gradients
    model2, g2 = load_model(z)
    loss2, out3 = ops.cross_entropy(model2(g2), dev(z["y"]), cw)
    loss2.backward()
    assert abs(loss2.item() - float(z["loss"])) < 1e-5
    for (k, p), (_, q) in zip(model.named_parameters(), model2.named_parameters()):
        np.testing.assert_allclose(q.grad.cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.usefixtures("both_gemm_modes")
def test_meansage_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "meansage_120.npz"))
    n, f0, hid, ncls, nl, _ = (int(v) for v in z["meta"])
    m = gte.MeanSAGE(f0, hid, ncls, nl)
    m.load_state_dict({k[len("state0."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state0.")})
    g = G.PageGraph(z["src"], z["dst"], n, device=DEV)
    with torch.no_grad():
        out = m.to(DEV)(g, dev(z["x"]), dev(z["w"])).cpu().numpy()
    np.testing.assert_allclose(out, z["out"], rtol=1e-5, atol=1e-5)


def test_meansage_backward_matches_the_oracles_autograd():
    """MeanSAGE (models.py:118-170; never instantiated by the reference's scripts, golden forward above): every parameter
    gradient and the input gradient of sum(out * up) against the CPU oracle's autograd on the same weights -- 1e-4 of each
    tensor's largest entry."""
    z = np.load(os.path.join(GOLDEN_DIR, "meansage_120.npz"))
    n, f0, hid, ncls, nl, _ = (int(v) for v in z["meta"])
    m = gte.MeanSAGE(f0, hid, ncls, nl)
    m.load_state_dict({k[len("state0."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state0.")})
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = sorted({k.rsplit(".", 1)[0] for k in sd})            # one prefix per linear
    order = sorted(names, key=lambda s: [int(t) for t in s.split(".") if t.isdigit()])
    ws = [(sd[p + ".weight"].clone().requires_grad_(True), sd[p + ".bias"].clone().requires_grad_(True)) for p in order]
    xo = torch.from_numpy(z["x"]).clone().requires_grad_(True)
    up = torch.randn(n, ncls, generator=torch.Generator().manual_seed(9))
    og = oc.OracleGraph(z["src"], z["dst"], n, z["w"])
    want = oc.meansage_forward(ws, og, xo)
    np.testing.assert_allclose(want.detach().numpy(), z["out"], rtol=1e-5, atol=1e-5)   # the oracle reproduces the golden forward
    (want * up).sum().backward()
    m = m.to(DEV)
    g = G.PageGraph(z["src"], z["dst"], n, device=DEV)
    xd = dev(z["x"]).requires_grad_(True)
    out = m(g, xd, dev(z["w"]))
    (out * up.to(DEV)).sum().backward()
    got = dict(m.named_parameters())
    for p, (w, b) in zip(order, ws):
        for name, ref in ((p + ".weight", w.grad), (p + ".bias", b.grad)):
            r = ref.numpy()
            np.testing.assert_allclose(got[name].grad.cpu().numpy(), r, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(r).max())
    r = xo.grad.numpy()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), r, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(r).max())


@pytest.mark.usefixtures("both_gemm_modes")
def test_batched_pages_vs_oracle_and_per_page_equivalence():
    """100-page batch (BASELINE cfg2 shape, F0=13): forward vs the CPU oracle; and batching must not
    change a page's logits (block-diagonal: no edge crosses pages)."""
    pages = S.make_pages(100, in_feats=13)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    torch.manual_seed(0)
    model = gte.GcnSAGE(13, 256, 9, 3, torch.nn.functional.relu, 0)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    want = oc.gcnsage_forward(state, oc.OracleGraph(src, dst, int(off[-1]), w), torch.from_numpy(feat)).numpy()
    model = model.to(DEV)
    gs = []
    for p in pages:
        g = G.PageGraph(p.src, p.dst, p.num_nodes, device=DEV)
        g.ndata["feat"], g.edata["feat"] = dev(p.feat), dev(p.weight)
        gs.append(g)
    with torch.no_grad():
        got = model(G.batch(gs)).cpu().numpy()
        one = model(gs[7]).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(one, got[off[7]:off[8]])          # bitwise: same row-local summation order


def test_large_graph_linearity_and_permutation_properties():
    """Size-independent properties at a size the CPU oracle would not finish quickly (200k nodes, F=512):
    linearity A(ax+by) = aA(x)+bA(y); all-ones input -> row sums of weights; mean of unit weights == 1."""
    rng = np.random.default_rng(11)
    n, deg, f = 200_000, 12, 512
    src = rng.integers(0, n, n * deg)
    dst = np.repeat(np.arange(n), deg)
    w = rng.random(n * deg).astype(np.float32)
    indptr, indices, perm, wout = ops.coo_to_csr(dev(dst, torch.int32), dev(src, torch.int32), n, dev(w))
    x, y = torch.randn(n, f, device=DEV), torch.randn(n, f, device=DEV)
    ax = ops.spmm_csr(indptr, indices, wout, x, n)
    ay = ops.spmm_csr(indptr, indices, wout, y, n)
    axy = ops.spmm_csr(indptr, indices, wout, 2.0 * x - 3.0 * y, n)
    assert torch.allclose(axy, 2.0 * ax - 3.0 * ay, rtol=1e-4, atol=1e-4)
    ones = ops.spmm_csr(indptr, indices, wout, torch.ones(n, 8, device=DEV), n)
    rowsum = torch.from_numpy(np.add.reduceat(w, np.arange(0, n * deg, deg))).to(DEV)
    assert torch.allclose(ones[:, 0], rowsum, rtol=1e-5, atol=1e-5)
    m1 = ops.spmm_csr(indptr, indices, None, torch.ones(n, 16, device=DEV), n, mean=True)
    assert torch.all(m1 == 1.0)


def test_cfg4_full_size_tiled_equals_plain_and_properties():
    """BASELINE cfg4 at its full size (1 M nodes, in-degree 12, F = 512 fp32, k-NN graph in Morton order): the LDS-staged
    kernel (two chunks in flight) against the row-per-wave kernel bit for bit, forward and on the reversed graph, plus the
    size-independent properties: mean of a constant, linearity."""
    n, k, f = 1_000_000, 12, 512
    src, dst, w = S.make_knn_stress_graph(n, k)
    x = torch.randn(n, f, device=DEV)
    for s_, d_ in ((src, dst), (dst, src)):                      # in-edge CSR (forward), out-edge CSR (backward)
        ip, ix, perm, wt = ops.coo_to_csr(dev(d_, torch.int32), dev(s_, torch.int32), n, dev(w))
        plan = ops.build_tile_plan(ip, ix, n)
        a = ops.spmm_csr(ip, ix, wt, x, n, mean=True)
        b = ops.spmm_csr(ip, ix, wt, x, n, mean=True, tiles=plan, force_tiled=True)
        assert torch.equal(a, b)
        del a
        c = ops.spmm_csr(ip, ix, None, torch.full((n, 128), 3.0, device=DEV), n, mean=True, tiles=plan, force_tiled=True)
        deg = (ip[1:] - ip[:-1]).unsqueeze(1)
        assert torch.allclose(c, torch.where(deg > 0, 3.0, 0.0).expand(n, 128), rtol=1e-6, atol=0)   # mean of a constant
        y = torch.randn(n, f, device=DEV)
        by = ops.spmm_csr(ip, ix, wt, y, n, mean=True, tiles=plan, force_tiled=True)
        bxy = ops.spmm_csr(ip, ix, wt, 2.0 * x - 3.0 * y, n, mean=True, tiles=plan, force_tiled=True)
        assert torch.allclose(bxy, 2.0 * b - 3.0 * by, rtol=1e-4, atol=1e-4)
        del b, by, bxy, y, c


# ---------------------------------------------------------------- hand-scheduled step engine
@pytest.mark.usefixtures("both_gemm_modes")
@pytest.mark.parametrize("name", ["page200_f13_l3_cw", "page300_f831_l3", "batch5_hetero", "single_node", "tiny_6n_10e"])
def test_fused_step_matches_reference_golden_and_autograd_path(name):
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep, TrainStep
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cw = dev(z["class_weights"]) if "class_weights" in z.files else None
    model, g = load_model(z)
    fused = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4, class_weights=cw)
    out3 = fused.forward_backward(g, dev(z["y"]).float())          # float labels, as loader.py:350-354 stores them
    assert abs(float(out3[0]) - float(z["loss"])) < 1e-5
    for k, p in model.named_parameters():
        ref = z["grad." + k]
        got = fused._gslice[id(p)].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(ref).max())
    model2, g2 = load_model(z)
    auto = TrainStep(model2, lr=0.01, weight_decay=5e-4, class_weights=cw)
    auto.step(g2, dev(z["y"]))
    p0 = fused.flat_param.detach().cpu().numpy().copy()
    fused.t += 1
    fused._optimizer_step()
    # The two paths sum in different orders (q-form / transform-first vs aggregate-first).  The first Adam step maps a
    # (weight-decayed) gradient g to lr * g / (|g| + eps): well conditioned except where g ~ eps, where a last-bit
    # difference moves the update by a visible fraction of lr.  Everything else must agree to 1e-5.
    got, want = fused.flat_param.cpu().numpy(), auto.flat_param.cpu().numpy()
    bad = ~np.isclose(got, want, rtol=1e-5, atol=1e-5)
    g_eff = np.abs(fused.flat_grad.cpu().numpy() + 5e-4 * p0)
    assert bad.mean() < 2e-4 and (g_eff[bad] < 1e-5).all() and np.abs(got - want).max() <= 0.02
    with torch.no_grad():
        after = model(g).cpu().numpy()
    poststep.check_against_fixture(z, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, after)


@pytest.mark.usefixtures("both_gemm_modes")
def test_fused_step_graph_replay_is_bitwise_the_eager_step():
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    pages = S.make_pages(12, in_feats=63)
    src, dst, w, feat, label, off = S.concat_pages(pages)

    def fresh():
        torch.manual_seed(1)
        m = gte.GcnSAGE(63, 96, 9, 3, torch.nn.functional.relu, 0).to(DEV)
        g = G.PageGraph(src, dst, int(off[-1]), device=DEV)
        g.ndata["feat"], g.edata["feat"] = dev(feat), dev(w)
        return FusedGcnSageStep(m, lr=0.01, weight_decay=5e-4), g

    a, ga = fresh()
    b, gb = fresh()
    y = dev(label)
    replay = b.capture(gb, y)
    for i in range(5):
        if i == 3:                                             # Adam sits inside the graph: lr is read from device memory
            a.lr_scale(0.5)
            b.lr_scale(0.5)
        la = a.step(ga, y).clone()
        lb = replay().clone()
        assert torch.equal(la, lb)
    assert torch.equal(a.flat_param, b.flat_param) and a.t == b.t == 5
    assert float(la[0]) < 2.2                                  # the loss goes down from ln(9)


# ---------------------------------------------------------------- LDS-staged (tiled) aggregation
@pytest.mark.parametrize("f", [32, 63, 64, 96, 100, 128, 160, 256, 512, 831])
@pytest.mark.parametrize("mean", [False, True])
def test_spmm_tiled_is_bitwise_the_plain_kernel(f, mean):
    """Same CSR order => the LDS-staged kernel must reproduce the row-per-wave kernel bit for bit,
    on a page-like graph (locality), with hub rows (direct-gather tiles) and an empty row."""
    pages = S.make_pages(40, in_feats=13)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    n = int(off[-1])
    rng = np.random.default_rng(f)
    hub_src = rng.integers(0, n, 3000)                         # row 100: 3000 extra in-edges (> EMAX)
    far_src = rng.integers(0, n, 40 * 30)                      # rows 2000..2029: 40 random far sources each (> UMAX)
    src = np.concatenate([src, hub_src, far_src])
    dst = np.concatenate([dst, np.full(3000, 100), np.repeat(np.arange(2000, 2030), 40)])
    w = np.concatenate([w, rng.random(len(src) - len(w)).astype(np.float32)])
    g = oc.OracleGraph(src, dst, n, w)
    ip, ix, wt = dev(g.indptr), dev(g.indices), dev(g.weight)
    x = dev(rng.standard_normal((n, f)).astype(np.float32))
    plan = ops.build_tile_plan(ip, ix, n)
    assert plan.max_unique > 128                                # both the staged and the direct path are exercised
    a = ops.spmm_csr(ip, ix, wt, x, n, mean=mean)
    b = ops.spmm_csr(ip, ix, wt, x, n, mean=mean, tiles=plan, force_tiled=True)
    assert torch.equal(a, b)
    c = ops.spmm_csr(ip, ix, None, x, n, mean=mean, tiles=plan, force_tiled=True)
    assert torch.equal(c, ops.spmm_csr(ip, ix, None, x, n, mean=mean))
    base = torch.randn(n, f, device=DEV)
    o1, o2 = base.clone(), base.clone()
    ops.spmm_csr(ip, ix, wt, x, n, mean=mean, out=o1, accumulate=True)
    ops.spmm_csr(ip, ix, wt, x, n, mean=mean, out=o2, accumulate=True, tiles=plan, force_tiled=True)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("seed", range(6))
def test_spmm_tiled_full_chunk_kernel_randomised(seed):
    """spmm_tiled_full_kernel (n_feat % 32 == 0: two register sets of hand-counted hidden loads in flight) against the
    row-per-wave kernel, bit for bit, over random widths, page counts, strided inputs/outputs and repeated launches
    (a wait released one load early would show up as run-to-run differences)."""
    rng = np.random.default_rng(1000 + seed)
    pages = S.make_pages(int(rng.integers(5, 60)), in_feats=13, first_id=int(rng.integers(0, 10_000)))
    src, dst, w, feat, label, off = S.concat_pages(pages)
    n = int(off[-1])
    # even seeds: the in-edge CSR (equal degrees inside most tiles); odd seeds: the out-edge CSR (degrees 0 ... ~30 inside a
    # tile: the kernel hands the tile's rows to its lane groups in order of decreasing degree)
    g = oc.OracleGraph(src, dst, n, w) if seed % 2 == 0 else oc.OracleGraph(dst, src, n, w)
    ip, ix, wt = dev(g.indptr), dev(g.indices), dev(g.weight)
    plan = ops.build_tile_plan(ip, ix, n)
    for f in rng.choice(np.arange(1, 17) * 32, size=4, replace=False):
        f = int(f)
        pad = int(rng.choice([0, 4, 8, 1, 3]))                       # row strides that are / are not a multiple of 16 bytes
        xs = torch.randn(n, f + pad, device=DEV)
        x = xs[:, :f]                                              # row stride > width
        a = ops.spmm_csr(ip, ix, wt, x, n, mean=True)
        outs = torch.zeros(n, f + pad, device=DEV)
        for rep in range(3):
            b = ops.spmm_csr(ip, ix, wt, x, n, mean=True, out=outs[:, :f], tiles=plan, force_tiled=True)
            assert torch.equal(a, b), (f, pad, rep)
        assert float(outs[:, f:].abs().sum()) == 0.0               # nothing written past the row
        base = torch.randn(n, f, device=DEV)
        o1, o2 = base.clone(), base.clone()
        ops.spmm_csr(ip, ix, wt, x, n, mean=False, out=o1, accumulate=True)
        ops.spmm_csr(ip, ix, wt, x, n, mean=False, out=o2, accumulate=True, tiles=plan, force_tiled=True)
        assert torch.equal(o1, o2), (f, pad)


def test_tile_plan_contract_and_oracle_parity():
    pages = S.make_pages(30, in_feats=13)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    n = int(off[-1])
    g = oc.OracleGraph(src, dst, n, w)
    ip, ix = dev(g.indptr), dev(g.indices)
    plan = ops.build_tile_plan(ip, ix, n)
    R = plan.tile_rows
    tp, ts, li = plan.tile_ptr.cpu().numpy(), plan.tile_src.cpu().numpy(), plan.local_index.cpu().numpy()
    assert len(tp) == (n + R - 1) // R + 1 and tp[0] == 0 and tp[-1] == len(ts)
    for t in range(0, len(tp) - 1, 7):
        e0, e1 = g.indptr[t * R], g.indptr[min((t + 1) * R, n)]
        seg = ts[tp[t]:tp[t + 1]]
        np.testing.assert_array_equal(seg, np.unique(g.indices[e0:e1]))            # distinct, sorted
        np.testing.assert_array_equal(seg[li[e0:e1]], g.indices[e0:e1])            # local index round trip
    x = np.random.default_rng(0).standard_normal((n, 96)).astype(np.float32)
    want = oc.spmm_csr_numpy(g.indptr, g.indices, g.weight, x) * g.norm
    got = ops.spmm_csr(ip, ix, dev(g.weight), dev(x), n, mean=True, tiles=plan, force_tiled=True).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)


def test_model_with_and_without_tiles_is_bitwise_identical():
    pages = S.make_pages(40, in_feats=63)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    torch.manual_seed(3)
    model = gte.GcnSAGE(63, 128, 9, 3, torch.nn.functional.relu, 0).to(DEV)
    outs = []
    monkey = (G.PageGraph.TILES_MIN_NODES, ops.TILED_MIN_BYTES, ops.TILED_MIN_DEGREE)
    G.PageGraph.TILES_MIN_NODES, ops.TILED_MIN_BYTES, ops.TILED_MIN_DEGREE = 1, 1, 0      # force the tiled path at test size
    for use in (True, False):
        g = G.PageGraph(src, dst, int(off[-1]), device=DEV)
        g.use_tiles = use
        g.ndata["feat"], g.edata["feat"] = dev(feat), dev(w)
        logits = model(g)
        logits.square().mean().backward()
        outs.append((logits.detach().clone(), [p.grad.clone() for p in model.parameters()]))
        model.zero_grad()
        assert (g.in_tiles() is not None) == use
    G.PageGraph.TILES_MIN_NODES, ops.TILED_MIN_BYTES, ops.TILED_MIN_DEGREE = monkey
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


# ---------------------------------------------------------------- resident pages: device-side batching, edge weights
def _page_graphs(pages, device="cpu"):
    gs = []
    for p in pages:
        g = G.PageGraph(p.src, p.dst, p.num_nodes, device=device)
        g.ndata["feat"], g.ndata["label"] = torch.from_numpy(p.feat).to(device), torch.from_numpy(p.label.astype(np.float32)).to(device)
        g.edata["feat"] = torch.from_numpy(p.weight).to(device)
        gs.append(g)
    return gs


@pytest.mark.usefixtures("both_gemm_modes")
def test_resident_batch_is_bitwise_the_host_batch():
    pages = S.make_pages(25, in_feats=63)
    res = G.ResidentPages(_page_graphs(pages), DEV)
    assert len(res) == 25 and res.page_sizes() == [p.num_nodes for p in pages]
    ids = [7, 3, 19, 3, 0, 24]                                   # unordered, one page twice
    rb = res.batch(ids)
    hb = G.batch(_page_graphs([pages[i] for i in ids], DEV))
    w = hb.edata["feat"]
    for a, b in ((rb.in_csr(), hb.in_csr()), (rb.out_csr(), hb.out_csr())):
        assert torch.equal(a.indptr, b.indptr) and torch.equal(a.indices, b.indices)
    assert torch.equal(rb.in_weights(rb.edata["feat"]), hb.in_weights(w))
    assert torch.equal(rb.out_weights(rb.edata["feat"], True), hb.out_weights(w, True))
    assert torch.equal(rb.ndata["feat"], hb.ndata["feat"]) and torch.equal(rb.ndata["label"], hb.ndata["label"])
    assert rb.num_nodes() == hb.num_nodes() and rb.num_edges() == hb.num_edges()
    assert rb.batch_num_nodes().tolist() == [pages[i].num_nodes for i in ids]
    torch.manual_seed(0)
    model = gte.GcnSAGE(63, 64, 9, 3, torch.nn.functional.relu, 0).to(DEV)
    la = model(rb)
    lb = model(hb)
    assert torch.equal(la, lb)
    la.square().mean().backward()
    ga = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    lb.square().mean().backward()
    for x, y in zip(ga, model.parameters()):
        assert torch.equal(x, y.grad)
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    f1 = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    o1 = f1.forward_backward(rb, rb.ndata["label"]).clone()
    g1 = f1.flat_grad.clone()
    o2 = f1.forward_backward(hb, hb.ndata["label"]).clone()
    assert torch.equal(o1, o2) and torch.equal(g1, f1.flat_grad)


def test_edge_weights_kernel_matches_the_reference_formula_bitwise():
    pages = S.make_pages(12, in_feats=13)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    bbox = np.concatenate([p.bbox for p in pages]).astype(np.int32)
    gon = np.repeat(np.arange(len(pages)), [p.num_nodes for p in pages]).astype(np.int32)
    got = G.edge_weights_from_boxes(dev(bbox), dev(src), dev(dst), dev(gon), len(pages)).cpu().numpy()
    np.testing.assert_array_equal(got, w)                         # the generator applies the host formula in float64
    # a page whose edges all have distance 0 (overlapping boxes): weight 1, no division by zero
    b = np.array([[0, 0, 10, 10], [5, 5, 15, 15]], dtype=np.int32)
    one = G.edge_weights_from_boxes(dev(b), dev(np.array([0, 1])), dev(np.array([1, 0])), dev(np.zeros(2, np.int32)), 1)
    assert one.cpu().tolist() == [1.0, 1.0]


def test_fused_step_buffers_do_not_grow_with_distinct_batch_sizes():
    """In the real loop every batch has a different node count: buffers are capacity-based row views, not per-count."""
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    pages = S.make_pages(60, in_feats=13)
    res = G.ResidentPages(_page_graphs(pages), DEV)
    torch.manual_seed(0)
    model = gte.GcnSAGE(13, 64, 9, 3, torch.nn.functional.relu, 0).to(DEV)
    eng = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    rng = np.random.default_rng(0)
    big = res.batch(list(range(40)))
    eng.step(big, big.ndata["label"])
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    sizes = set()
    for _ in range(25):
        ids = rng.choice(60, size=int(rng.integers(5, 35)), replace=False)
        bg = res.batch(ids)
        sizes.add(bg.num_nodes())
        out3 = eng.step(bg, bg.ndata["label"])
    torch.cuda.synchronize()
    assert len(sizes) > 15 and len(eng._bufs) == 1
    assert torch.cuda.memory_allocated() - base < 8 << 20        # only the (freed) per-batch graphs' worth of slack
    assert np.isfinite(float(out3[0]))


def test_adam_inside_the_fold_launch_is_bitwise_the_separate_launch():
    """gte_fold_defer_flush_adam: when the deferred folds produce every gradient element (the cfg2 step: split-K dW, LayerNorm
    column sums, the narrow layer's dW) the optimiser step rides in the fold launch.  Same parameters, moments and step
    count, bit for bit, as fold launch + gte_adam_step_dev; a model whose dW GEMM writes its gradient directly (few nodes:
    no split-K) falls back to the separate launch."""
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    pages = S.make_pages(40, in_feats=200)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    g = G.PageGraph(src, dst, int(off[-1]), device=DEV)
    g.ndata["feat"], g.edata["feat"] = dev(feat), dev(w)
    y = dev(label).float()
    states = {}
    for fuse in (True, False):
        torch.manual_seed(3)
        model = gte.GcnSAGE(200, 256, 9, 3, torch.nn.functional.relu, 0).to(DEV)
        eng = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
        eng.fuse_adam = fuse
        for _ in range(4):
            eng.step(g, y)
        torch.cuda.synchronize()
        assert eng.adam_fused_steps == (4 if fuse else 0)
        assert int(eng._step_dev.item()) == 4
        states[fuse] = (eng.flat_param.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), eng.flat_grad.clone(), eng._hyper.clone())
    for a, b in zip(states[True], states[False]):
        assert torch.equal(a, b)
    # a graph too small for split-K: some gradients are written by their GEMM directly -> no fusion, same results as ever
    small = S.make_pages(1, in_feats=200)
    src, dst, w, feat, label, off = S.concat_pages(small)
    gs = G.PageGraph(src, dst, int(off[-1]), device=DEV)
    gs.ndata["feat"], gs.edata["feat"] = dev(feat), dev(w)
    torch.manual_seed(3)
    model = gte.GcnSAGE(200, 256, 9, 3, torch.nn.functional.relu, 0).to(DEV)
    eng = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    eng.step(gs, dev(label).float())
    torch.cuda.synchronize()
    assert int(eng._step_dev.item()) == 1 and bool(torch.isfinite(eng.flat_param).all())


# ---------------------------------------------------------------- the headline model at full width, whole model
@pytest.mark.usefixtures("both_gemm_modes")
def test_headline_shape_case_matches_reference_golden():
    """GcnSAGE(831, 256, 9, 3) on a 2 000-node graph against the reference's own vectors (trimmed fixture):
    forward through the module path at 1e-5, then ONE fused step (transform-first / q-form / fused head): loss 1e-5,
    gradients 1e-4, parameters and post-step logits per tests/poststep.py."""
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    z, src, dst, w, x, y, state0, model = poststep.headline_case(GOLDEN_DIR)
    model = model.to(DEV)
    g = G.PageGraph(src, dst, len(x), device=DEV)
    g.ndata["feat"], g.edata["feat"] = dev(x), dev(w)
    hidden = []
    hooks = [l.register_forward_hook(lambda m, i, o: hidden.append(o.detach().cpu().numpy())) for l in model.layers]
    with torch.no_grad():
        logits = model(g).cpu().numpy()
    for h in hooks:
        h.remove()
    fused = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    out3 = fused.step(g, dev(y).float())
    grads = {k: fused._gslice[id(p)].cpu().numpy() for k, p in model.named_parameters()}
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    with torch.no_grad():
        after = model(g).cpu().numpy()
    og = oc.OracleGraph(src, dst, len(x), w)
    ref_after = oc.gcnsage_forward({k: torch.from_numpy(v) for k, v in params.items()}, og, torch.from_numpy(x)).numpy()
    poststep.check_headline(z, logits, hidden[:3], float(out3[0]), grads, params, after, state0, oracle_after=ref_after)


@pytest.mark.usefixtures("both_gemm_modes")
def test_cfg2_primary_full_size_step_matches_the_oracle():
    """BASELINE configs[1] at FULL size -- 100 pages (~24.5 k nodes), F0 = 831, hidden 256, 3 layers -- as ONE model: the fused
    step of the train loop against the CPU oracle's step (OracleTrainer, itself pinned to the reference's golden vectors):
    forward logits 1e-5, loss 1e-5, every gradient 1e-4, parameters after the step per tests/poststep.py, post-step logits 1e-4."""
    from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
    pages = S.make_pages(100, in_feats=831)
    src, dst, w, feat, label, off = S.concat_pages(pages)
    n = int(off[-1])
    torch.manual_seed(42)
    model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    og = oc.OracleGraph(src, dst, n, w)
    xt, yt = torch.from_numpy(feat), torch.from_numpy(label)
    want_logits = oc.gcnsage_forward(state0, og, xt).numpy()
    tr = oc.OracleTrainer(state0, lr=0.01, weight_decay=5e-4)
    want_loss, _ = tr.step(og, xt, yt)
    want_grads = {k: v.numpy() for k, v in tr.grads().items()}
    want_state = {k: v.detach().numpy() for k, v in tr.state.items()}

    model = model.to(DEV)
    g = G.PageGraph(src, dst, n, device=DEV)
    g.ndata["feat"], g.edata["feat"] = dev(feat), dev(w)
    with torch.no_grad():
        logits = model(g).cpu().numpy()
    np.testing.assert_allclose(logits, want_logits, rtol=1e-5, atol=1e-5)
    fused = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
    out3 = fused.step(g, dev(label).float())
    assert abs(float(out3[0]) - want_loss) < 1e-5
    for k, p in model.named_parameters():
        got, ref = fused._gslice[id(p)].cpu().numpy(), want_grads[k]
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(ref).max())
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    g_eff = {k: np.abs(want_grads.get(k, np.zeros_like(v)) + 5e-4 * state0[k].numpy()) for k, v in want_state.items()}
    hyb = poststep.hybrid_state(want_state, params, g_eff)
    with torch.no_grad():
        after = model(g).cpu().numpy()
    ref_after = oc.gcnsage_forward(hyb, og, xt).numpy()
    assert np.abs(after - ref_after).max() < 1e-4


@pytest.mark.parametrize("p_drop", [0.3, 0.5])
def test_dropout_layer_draws_one_mask_over_the_concatenation_and_matches_torch(p_drop):
    """GcnSAGELayer with dropout > 0 in training mode (models.py:58-66 of the reference: `h = concat(h, ah * norm)`, then ONE
    `self.dropout(h)` over [N, 2 in], linear, LayerNorm, ReLU): with the device generator re-seeded, the layer's output and its
    gradients equal the same formula in torch with the mask that one call draws (the reference's RNG consumption: one draw per
    layer over the concatenation; round 4 drew two)."""
    rng = np.random.default_rng(11)
    n, f, out = 300, 24, 40
    src, dst = rng.integers(0, n, 5 * n), rng.integers(0, n, 5 * n)
    w = rng.uniform(0.1, 1.0, 5 * n).astype(np.float32)
    x = rng.standard_normal((n, f)).astype(np.float32)
    g = gte.PageGraph(src, dst, n, device=DEV)
    g.edata["feat"] = dev(w)
    torch.manual_seed(3)
    layer = gte.GcnSAGELayer(f, out, torch.nn.functional.relu, p_drop).to(DEV)
    layer.train()
    xh = dev(x).requires_grad_(True)
    torch.manual_seed(77)
    y = layer(g, xh)
    y.square().sum().backward()
    gW, gx = layer.linear.weight.grad.clone(), xh.grad.clone()
    # the same in torch: the mask of ONE dropout call over [n, 2 f] from the same generator state
    og = oc.OracleGraph(src, dst, n, w)
    torch.manual_seed(77)
    mask = torch.nn.functional.dropout(torch.ones((n, 2 * f), device=DEV), p_drop, training=True)
    xr = dev(x).double().requires_grad_(True)
    A = torch.zeros((n, n), dtype=torch.float64, device=DEV)
    A.index_put_((dev(dst).long(), dev(src).long()), dev(w).double(), accumulate=True)
    norm = dev(og.norm).double()
    cat = torch.cat((xr, (A @ xr) * norm), dim=1) * mask.double()
    W, b = layer.linear.weight.detach().double().requires_grad_(True), layer.linear.bias.detach().double()
    z = cat @ W.t() + b
    ref = torch.relu(torch.nn.functional.layer_norm(z, (out,), layer.lynorm.weight.detach().double(), layer.lynorm.bias.detach().double(), 1e-5))
    ref.square().sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gW.cpu().numpy(), W.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(W.grad.abs().max()))
    np.testing.assert_allclose(gx.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(xr.grad.abs().max()))
    layer.eval()                                                  # evaluation mode: no mask
    with torch.no_grad():
        y_eval = layer(g, dev(x))
    assert not torch.equal(y_eval, y.detach())
