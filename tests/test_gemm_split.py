"""The split-bf16 arithmetic mode of the transform GEMMs (csrc/gemm_split.h, gte_gemm_set_mode): fp32 operands cut exactly
into three bf16 pieces, six bf16 MFMA partial products, fp32 accumulation.

What is checked, on an MI355X (``-m gpu``), through the C ABI:
  * against fp64, in units of u = 2^-24 * sum_k |a_k b_k| (one fp32 rounding at the magnitude of the dot product): the split
    kernel's error is not above 1.25 x the fp32 MFMA kernel's + 2.5 u and below an absolute 12 u on every operand layout,
    ragged shape, K tail, two-segment K and split-K weight gradient.  (The 2.5 u: one product's dropped terms ml + lm are
    bounded by 2 u -- a single correctly rounded fp32 product is within 1 u -- so at K = 1 the split kernel shows up to
    ~2.5 u against the fp32 kernel's 1 u; from K ~ 16 on the accumulation roundings dominate and the split kernel's error is
    the smaller one: it rounds once per 16-term MFMA block, the fp32 kernel once per 2-term block.);
  * the whole-model parity tests of test_gpu_parity.py (reference golden vectors at the headline width; cfg2 at full size
    against the oracle's step) pass UNCHANGED in this mode -- forward 1e-5, loss 1e-5, gradients 1e-4;
  * the tail split and the epilogue flags (bias / relu / accumulate) keep their meaning; the mode switch validates its input.
"""
import numpy as np
import pytest
import torch

import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def split_mode():
    prev = ops.set_gemm_mode("split_bf16")
    try:
        yield
    finally:
        ops.set_gemm_mode(prev)


def _both_modes(fn):
    out = {}
    for mode in ("f32", "split_bf16"):
        prev = ops.set_gemm_mode(mode)
        try:
            out[mode] = fn()
        finally:
            ops.set_gemm_mode(prev)
    return out


def _err_units(c, ref, unit):
    return float(((c.double().cpu() - ref).abs() / unit.clamp_min(1e-300)).max())


def _check(res_units):
    assert res_units["split_bf16"] <= 1.25 * res_units["f32"] + 2.5, res_units
    assert res_units["split_bf16"] < 12.0, res_units


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("m,n,k", [(3000, 512, 831), (1000, 256, 77), (300, 130, 1030), (129, 128, 16), (64, 128, 33),
                                   (640, 831, 512), (257, 200, 1), (128, 128, 4100)])
def test_split_gemm_error_against_fp64_not_above_the_fp32_kernel(ta, tb, m, n, k):
    g = torch.Generator().manual_seed(1000 * m + 10 * n + k)
    a = torch.randn((k, m) if ta else (m, k), generator=g) * torch.exp(torch.randn(m if ta else k, generator=g))
    b = torch.randn((n, k) if tb else (k, n), generator=g) * 0.1
    a64, b64 = (a.t() if ta else a).double(), (b.t() if tb else b).double()
    ref, unit = a64 @ b64, (a64.abs() @ b64.abs()) * 2.0 ** -24
    ad, bd = a.to(DEV), b.to(DEV)
    res = _both_modes(lambda: _err_units(ops.gemm(ad, bd, trans_a=ta, trans_b=tb), ref, unit))
    _check(res)


def test_split_gemm_randomised_shapes_strides_and_layouts():
    """60 random (M, N, K, layout) with operands and outputs that are column slices of wider buffers (leading dimensions that
    are not multiples of 4, unaligned row starts): the split kernel's windows, K tails, row tails and tile shapes."""
    rng = np.random.default_rng(20260)
    g = torch.Generator().manual_seed(9)
    for case in range(60):
        m = int(rng.choice([1, 31, 64, 65, 127, 128, 129, 200, 513, 1000, 2049]))
        n = int(rng.choice([33, 64, 100, 128, 129, 255, 256, 300, 512]))
        k = int(rng.choice([1, 3, 4, 15, 16, 17, 31, 33, 64, 100, 255, 256, 831, 1100]))
        ta, tb = bool(rng.integers(2)), bool(rng.integers(2))
        pa, pb, pc = int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4))     # column offsets
        wa, wb = int(rng.integers(0, 7)), int(rng.integers(0, 7))                                   # extra row width
        A = torch.randn((k, pa + m + wa) if ta else (m, pa + k + wa), generator=g).to(DEV)
        B = torch.randn((n, pb + k + wb) if tb else (k, pb + n + wb), generator=g).to(DEV)
        a = A[:, pa:pa + (m if ta else k)]
        b = B[:, pb:pb + (k if tb else n)]
        Cbuf = torch.full((m, pc + n + 3), 7.0, device=DEV)
        a64, b64 = (a.t() if ta else a).double().cpu(), (b.t() if tb else b).double().cpu()
        ref, unit = a64 @ b64, (a64.abs() @ b64.abs()) * 2.0 ** -24
        res = {}
        for mode in ("f32", "split_bf16"):
            prev = ops.set_gemm_mode(mode)
            try:
                Cbuf.fill_(7.0)
                ops.gemm(a, b, trans_a=ta, trans_b=tb, out=Cbuf[:, pc:pc + n])
            finally:
                ops.set_gemm_mode(prev)
            res[mode] = _err_units(Cbuf[:, pc:pc + n], ref, unit)
            assert bool((Cbuf[:, :pc] == 7.0).all()) and bool((Cbuf[:, pc + n:] == 7.0).all()), (case, mode, "wrote outside C")
        assert res["split_bf16"] <= 1.25 * res["f32"] + 2.5 and res["split_bf16"] < 12.0, (case, m, n, k, ta, tb, res)


def test_split_gemm_wide_dynamic_range_and_exact_cases():
    """Operands spanning 2^-60 .. 2^60 (the pieces keep fp32's exponent range); small integers come out exact."""
    g = torch.Generator().manual_seed(3)
    m, n, k = 512, 256, 300
    a = torch.randn(m, k, generator=g) * torch.exp2(torch.randint(-60, 60, (m, 1), generator=g).float())
    b = torch.randn(k, n, generator=g) * torch.exp2(torch.randint(-30, 30, (1, n), generator=g).float())
    ref, unit = a.double() @ b.double(), (a.double().abs() @ b.double().abs()) * 2.0 ** -24
    res = _both_modes(lambda: _err_units(ops.gemm(a.to(DEV), b.to(DEV)), ref, unit))
    _check(res)
    ai = torch.randint(-50, 50, (300, 200), generator=g).float()
    bi = torch.randint(-50, 50, (200, 260), generator=g).float()
    prev = ops.set_gemm_mode("split_bf16")
    try:
        got = ops.gemm(ai.to(DEV), bi.to(DEV)).cpu()
    finally:
        ops.set_gemm_mode(prev)
    assert torch.equal(got, ai @ bi)                     # every partial sum is an integer below 2^24


@pytest.mark.parametrize("nodes,n_out,k1,k2", [(6000, 256, 256, 256), (3000, 256, 100, 60), (9000, 128, 831, 0)])
def test_split_mode_weight_gradient_two_segments_split_k(nodes, n_out, k1, k2):
    """dW = dz^T [x1 | x2] (gte_sage_linear_dw): row-contiguous operands (transposed LDS reads), two K... N segments, split-K slabs."""
    g = torch.Generator().manual_seed(nodes + k1)
    dz = torch.randn(nodes, n_out, generator=g) * 0.01
    x1 = torch.randn(nodes, k1, generator=g)
    x2 = torch.randn(nodes, k2, generator=g) if k2 else None
    xx = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref, unit = dz.double().t() @ xx.double(), (dz.double().abs().t() @ xx.double().abs()) * 2.0 ** -24

    def run():
        o = torch.empty(n_out, k1 + k2, device=DEV)
        ops.sage_linear_dw(dz.to(DEV), x1.to(DEV), None if x2 is None else x2.to(DEV), o)
        return _err_units(o, ref, unit)
    _check(_both_modes(run))


def test_split_mode_linear_forward_two_k_segments_with_bias_and_layernorm():
    """y = relu(LN([a1 | a2] W^T + b)) (gte_sage_linear_fwd): two K segments with different leading dimensions."""
    g = torch.Generator().manual_seed(11)
    n, k1, k2, out = 2500, 300, 300, 256
    a1, a2 = torch.randn(n, k1, generator=g), torch.randn(n, k2, generator=g)
    w, b = torch.randn(out, k1 + k2, generator=g) * 0.05, torch.randn(out, generator=g)
    gamma, beta = torch.rand(out, generator=g) + 0.5, torch.randn(out, generator=g)
    z = torch.cat([a1, a2], 1).double() @ w.double().t() + b.double()
    want = torch.relu(torch.nn.functional.layer_norm(z, (out,), gamma.double(), beta.double(), 1e-5))

    def run():
        y, _, _ = ops.sage_linear_fwd(a1.to(DEV), a2.to(DEV), w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, True, False)
        return float((y.double().cpu() - want).abs().max())
    res = _both_modes(run)
    assert res["split_bf16"] < 1e-5 and res["split_bf16"] <= 1.5 * res["f32"] + 1e-6, res


def test_split_mode_tail_split_and_epilogue_flags(split_mode):
    """The tail split (K ranges of the last round's tiles) and accumulate keep their meaning in split mode."""
    lib = gte._lib.load()
    g = torch.Generator().manual_seed(5)
    m, n, k = 128 * 70, 512, 600                          # 280 tiles on 256 CUs: 24 tail tiles
    a, b = torch.randn(m, k, generator=g).to(DEV), torch.randn(n, k, generator=g).to(DEV)
    ref = a.double() @ b.double().t()
    plain = ops.gemm(a, b, trans_b=True)
    ws = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=DEV)
    gte._lib.check(lib.gte_gemm_set_tail_workspace(gte._lib.ptr(ws), ws.numel()), "set")
    try:
        tail = ops.gemm(a, b, trans_b=True)
        c0 = torch.randn(m, n, generator=g).to(DEV)
        acc = ops.gemm(a, b, trans_b=True, out=c0.clone(), accumulate=True)
    finally:
        lib.gte_gemm_set_tail_workspace(None, 0)
    scale = float(ref.abs().max())
    assert float((plain.double() - ref).abs().max()) < 2e-6 * scale
    assert float((tail.double() - ref).abs().max()) < 2e-6 * scale
    assert float((acc.double() - (ref + c0.double())).abs().max()) < 2e-6 * scale


def test_gemm_mode_switch_validates_and_round_trips():
    lib = gte._lib.load()
    prev = ops.get_gemm_mode()
    assert lib.gte_gemm_set_mode(7) != 0 and b"unknown mode" in lib.gte_last_error()
    assert ops.get_gemm_mode() == prev
    assert ops.set_gemm_mode("split_bf16") == prev and ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16
    ops.set_gemm_mode(prev)
    assert ops.get_gemm_mode() == prev


def test_thread_mode_override_wins_over_the_process_mode_for_this_thread_only():
    import threading
    lib = gte._lib.load()
    prev = ops.get_gemm_mode()
    assert lib.gte_gemm_set_thread_mode(5) != 0
    try:
        assert lib.gte_gemm_set_thread_mode(ops.GEMM_SPLIT_BF16) == 0 and ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16
        seen = []
        t = threading.Thread(target=lambda: seen.append(lib.gte_gemm_get_mode()))
        t.start(); t.join()
        assert seen == [prev]                               # another thread still sees the process-wide mode
    finally:
        lib.gte_gemm_set_thread_mode(-1)
    assert ops.get_gemm_mode() == prev


def test_headline_width_model_matches_the_reference_golden_in_split_mode(split_mode):
    from tests import test_gpu_parity as T
    T.test_headline_shape_case_matches_reference_golden()


def test_cfg2_full_size_step_matches_the_oracle_in_split_mode(split_mode):
    from tests import test_gpu_parity as T
    T.test_cfg2_primary_full_size_step_matches_the_oracle()


def _gcn_cases():
    from tests import test_gpu_parity as T
    return T.GCN_CASES


@pytest.mark.parametrize("name", _gcn_cases())
def test_reference_golden_cases_forward_and_train_step_in_split_mode(split_mode, name):
    """Every reference golden GCN case: forward (1e-5) and one train step through the module / autograd path."""
    from tests import test_gpu_parity as T
    T.test_gcnsage_forward_matches_reference_golden(name)
    T.test_gcnsage_train_step_matches_reference_golden(name)


@pytest.mark.parametrize("name", ["page200_f13_l3_cw", "page300_f831_l3", "batch5_hetero"])
def test_fused_step_matches_reference_golden_in_split_mode(split_mode, name):
    from tests import test_gpu_parity as T
    T.test_fused_step_matches_reference_golden_and_autograd_path(name)
