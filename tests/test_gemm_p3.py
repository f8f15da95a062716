"""The planes GEMMs (csrc/gemm_p3.hip) and the P3 operand format (csrc/p3.h): a fp32 matrix kept as three bf16 planes made ONCE by
its producer, multiplied as six bf16 MFMA partial products with fp32 accumulation -- the arithmetic of the split mode
(tests/test_gemm_split.py) without the per-stage splitting.

Checked on an MI355X through the C ABI:
  * P3 round trip is exact (h + m + l == x bit for bit), transposed too, ragged shapes;
  * NT (forward transform / dX) is BIT-IDENTICAL to the split kernel on the fp32 operands, on every tile configuration the chooser
    can pick and on the ones it never picks (GTE_P3_NT_CFG);
  * TN (dW, split-K) against fp64 in units of u = 2^-24 sum_k |a_k b_k|, not above the split kernel's;
  * small-integer operands come out exact; bias / relu / accumulate keep their meaning;
  * the ROW-MAP forms (gte_gemm_p3_nt_rows / gte_gemm_p3_tn_rows: the operand is a subset of the rows of a resident image) are
    bit-identical to the GEMM on a gathered copy of those rows;
  * non-finite and denormal-range operands: see the two tests at the end for what the six-product arithmetic guarantees.
"""
import numpy as np
import pytest
import torch

from gnn_tableextraction_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def split_mode():
    prev = ops.set_gemm_mode("split_bf16")
    try:
        yield
    finally:
        ops.set_gemm_mode(prev)


def _wide(r, c, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(r, c, device=DEV, generator=g) * torch.exp(3 * torch.randn(r, c, device=DEV, generator=g))


def _weights(img, block_major):
    """a row-major weight image in the layout under test (block-major: ops.P3 -- block fb is one contiguous [rows][96 bytes] run)"""
    if not block_major:
        return img
    r, ldp = img.data.shape
    return ops.P3(img.data.view(r, ldp // 96, 96).permute(1, 0, 2).contiguous(), img.rows, img.cols, block_major=True)


@pytest.fixture(params=[False, True], ids=["w_row_major", "w_block_major"])
def wlayout(request):
    return request.param


def _units(c, ref64, unit):
    return float(((c.double() - ref64).abs() / unit.clamp_min(1e-300)).max())


@pytest.mark.parametrize("r,c", [(5, 3), (300, 831), (1000, 256), (77, 16), (64, 17), (1, 1)])
def test_p3_round_trip_is_exact(r, c):
    x = _wide(r, c)
    img = ops.p3_from_f32(x)
    assert img.ldp == _lib.load().gte_p3_row_bytes(c) == 96 * ((c + 15) // 16)
    assert torch.equal(ops.p3_to_f32(img), x)
    assert torch.equal(ops.p3_to_f32(ops.p3_from_f32(x, transpose=True)), x.t())
    # the block-major layout (weight images): the same 96-byte blocks, block fb of every row in one run
    blk = ops.p3_from_f32(x, block_major=True)
    assert blk.ldp == -96 * r and torch.equal(blk.data, _weights(img, True).data)
    assert torch.equal(ops.p3_to_f32(blk), x)
    assert torch.equal(ops.p3_to_f32(ops.p3_from_f32(x, transpose=True, block_major=True)), x.t())


@pytest.mark.parametrize("m,n,k1,k2", [(500, 256, 256, 256), (24437, 256, 831, 831), (24437, 512, 256, 0), (333, 100, 40, 24), (3000, 200, 363, 363),
                                       (4100, 48, 1030, 0), (2000, 1000, 256, 256)])
def test_block_major_weights_give_the_same_bits(split_mode, m, n, k1, k2):
    """every NT planes GEMM takes its B operand row-major or block-major (a negative ldp at the ABI): same products, same bits"""
    g = torch.Generator(device=DEV).manual_seed(m + n + k1)
    kp = -(-k1 // 16) * 16
    a1 = ops.p3_from_f32(torch.randn(m, k1, device=DEV, generator=g))
    a2 = ops.p3_from_f32(torch.randn(m, k2, device=DEV, generator=g)) if k2 else None
    wb = torch.zeros(n, kp + k2, device=DEV)
    wb[:, :k1] = torch.randn(n, k1, device=DEV, generator=g)
    if k2:
        wb[:, kp:] = torch.randn(n, k2, device=DEV, generator=g)
    w = ops.p3_from_f32(wb)
    bias = torch.randn(n, device=DEV, generator=g)
    want = ops.gemm_p3_nt(a1, w, a2=a2, bias=bias)
    assert torch.equal(ops.gemm_p3_nt(a1, _weights(w, True), a2=a2, bias=bias), want)
    # ... written in place, half by half, as the engine's weight arrangements are (ops.P3.at)
    img = ops.P3.empty(n, kp + k2, DEV, block_major=True)
    img.data.zero_()
    ops.p3_from_f32(wb[:, :kp].contiguous(), out=img)
    if k2:
        ops.p3_from_f32(wb[:, kp:].contiguous(), out=img, block0=kp // 16)
    assert torch.equal(img.data, _weights(w, True).data)


@pytest.mark.parametrize("m,n,k", [(100, 128, 16), (128, 128, 32), (300, 512, 831), (2000, 256, 256), (24437, 512, 831),
                                   (24437, 512, 256), (777, 200, 50), (1, 16, 7), (4100, 48, 1030)])
def test_nt_is_bitwise_the_split_kernel(split_mode, m, n, k):
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    a, b = torch.randn(m, k, device=DEV, generator=g), torch.randn(n, k, device=DEV, generator=g)
    bias = torch.randn(n, device=DEV, generator=g)
    ap, bp = ops.p3_from_f32(a), ops.p3_from_f32(b)
    got = ops.gemm_p3_nt(ap, bp)
    want = ops.gemm(a, b, trans_b=True)
    if (m, n, k) in ((300, 512, 831), (1, 16, 7), (4100, 48, 1030)):
        # few tiles and a long K: the split kernel cuts K into slabs here (another summation order); a one-row product takes its
        # small-problem path: same accuracy, other bits
        ref, unit = a.double() @ b.double().t(), 2.0 ** -24 * (a.double().abs() @ b.double().abs().t())
        assert _units(got, ref, unit) < 6.0 and _units(want, ref, unit) < 6.0
        want = got.clone()
    assert torch.equal(got, want)
    # epilogue: bias on the first bias_cols columns, relu, accumulate
    half = n // 2
    gb = ops.gemm_p3_nt(ap, bp, bias=bias, bias_cols=half, relu=True)
    wb = (want + torch.cat([bias[:half], torch.zeros(n - half, device=DEV)])).clamp_min(0)
    assert torch.equal(gb, wb)
    acc = torch.full((m, n), 0.5, device=DEV)
    ops.gemm_p3_nt(ap, bp, out=acc, accumulate=True)
    assert torch.equal(acc, want + 0.5)


ALL_NT_CFGS = [0, 1, 2, 3, 4, 5, 6, 7]


@pytest.fixture(params=[0, 32, 64, 96, 128], ids=lambda r: f"ln_rows{r}" if r else "ln_rows_auto")
def ln_rows(request):
    """every row tile of the launches with a LayerNorm epilogue (gte_gemm_p3_set_ln_rows; 0 = the chooser), the chooser again afterwards"""
    lib = _lib.load()
    assert lib.gte_gemm_p3_set_ln_rows(int(request.param)) == 0
    yield request.param
    assert lib.gte_gemm_p3_set_ln_rows(0) == 0


@pytest.fixture
def force_nt_cfg():
    """force a tile configuration of the NT planes GEMM for the test (gte_gemm_p3_set_nt_cfg), the chooser again afterwards"""
    lib = _lib.load()

    def force(cfg):
        assert lib.gte_gemm_p3_set_nt_cfg(-1 if cfg is None else int(cfg)) == 0
    yield force
    assert lib.gte_gemm_p3_set_nt_cfg(-1) == 0


@pytest.mark.parametrize("cfg", ALL_NT_CFGS)
@pytest.mark.parametrize("n", [500, 192, 200, 448])
def test_every_nt_tile_configuration_gives_the_same_bits(split_mode, force_nt_cfg, cfg, n):
    """The chooser picks one tile shape per problem; each shape, forced, gives the bits of the split kernel (ragged M and N; output
    widths of the reference's scaled runs: 2 x 96, 2 x 100, 2 x 224)."""
    g = torch.Generator(device=DEV).manual_seed(cfg)
    a, b = torch.randn(3001, 363, device=DEV, generator=g), torch.randn(n, 363, device=DEV, generator=g)
    want = ops.gemm(a, b, trans_b=True)
    force_nt_cfg(cfg)
    assert torch.equal(ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b)), want)


@pytest.mark.parametrize("m,n,k1,k2", [(500, 256, 256, 256), (24437, 256, 256, 256), (333, 100, 40, 24)])
def test_nt_two_k_segments(m, n, k1, k2):
    """dX = dz W_s + q W_n: A = [a1 | a2], segment 1 padded to whole 16-feature blocks in b."""
    g = torch.Generator(device=DEV).manual_seed(k1)
    a1, a2 = torch.randn(m, k1, device=DEV, generator=g), torch.randn(m, k2, device=DEV, generator=g)
    w = torch.randn(n, k1 + k2, device=DEV, generator=g)
    kb1 = -(-k1 // 16) * 16
    full = torch.zeros(n, kb1 + k2, device=DEV)
    full[:, :k1], full[:, kb1:] = w[:, :k1], w[:, k1:]
    c = ops.gemm_p3_nt(ops.p3_from_f32(a1), ops.p3_from_f32(full), a2=ops.p3_from_f32(a2))
    cat = torch.cat([a1, a2], 1).double()
    ref, unit = cat @ w.double().t(), 2.0 ** -24 * (cat.abs() @ w.double().abs().t())
    assert _units(c, ref, unit) < 6.0


@pytest.mark.parametrize("m,n,k,two", [(256, 256, 64, False), (256, 831, 1000, False), (256, 256, 24437, True),
                                       (256, 831, 24437, True), (100, 50, 333, False), (218, 63, 5000, True), (128, 13, 17, True)])
def test_tn_against_fp64(split_mode, m, n, k, two):
    g = torch.Generator(device=DEV).manual_seed(k)
    a, b, a2 = (torch.randn(k, c, device=DEV, generator=g) for c in (m, n, m))
    if two:
        c = ops.gemm_p3_tn(ops.p3_from_f32(a), ops.p3_from_f32(b), a2=ops.p3_from_f32(a2), two_segments=True)
        ref = torch.cat([a.double().t() @ b.double(), a2.double().t() @ b.double()], 1)
        unit = 2.0 ** -24 * torch.cat([a.double().abs().t() @ b.double().abs(), a2.double().abs().t() @ b.double().abs()], 1)
    else:
        c = ops.gemm_p3_tn(ops.p3_from_f32(a), ops.p3_from_f32(b))
        ref, unit = a.double().t() @ b.double(), 2.0 ** -24 * (a.double().abs().t() @ b.double().abs())
    e = _units(c, ref, unit)
    e_split = _units(ops.gemm(a, b, trans_a=True), ref[:, :n], unit[:, :n])
    assert e < 12.0 and e <= 1.25 * e_split + 2.5, (e, e_split)


def test_small_integers_are_exact():
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randint(-8, 9, (1000, 100), device=DEV, generator=g).float()
    b = torch.randint(-8, 9, (300, 100), device=DEV, generator=g).float()
    assert torch.equal(ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b)).double(), a.double() @ b.double().t())
    a = torch.randint(-8, 9, (1003, 200), device=DEV, generator=g).float()
    b = torch.randint(-8, 9, (1003, 300), device=DEV, generator=g).float()
    assert torch.equal(ops.gemm_p3_tn(ops.p3_from_f32(a), ops.p3_from_f32(b)).double(), a.double().t() @ b.double())


# ---- row maps: the operand is a subset of the rows of a RESIDENT image -----------------------------------------------------
def _row_map(n_res, rows, seed):
    """page-like runs of consecutive resident rows, in shuffled order; padded as ResidentPages pads it"""
    rng = np.random.default_rng(seed)
    out, left = [], rows
    while left > 0:
        run = int(min(left, rng.integers(1, 400)))
        start = int(rng.integers(0, n_res - run + 1))
        out.append(np.arange(start, start + run))
        left -= run
    ids = np.concatenate(out).astype(np.int32)
    pad = np.full(33, n_res, dtype=np.int32)
    return torch.from_numpy(np.concatenate([ids, pad])).to(DEV)


def _resident_image(x):
    """P3 image of x with one more row, zero, behind it (what ResidentPages.enable_p3 allocates: the 64-bit row-map path reads
    the map's padding entries without a range check)"""
    img = ops.P3.empty(x.shape[0], x.shape[1], x.device, rows_cap=x.shape[0] + 1)
    img.data[x.shape[0]:].zero_()
    return ops.p3_from_f32(x, out=img)


@pytest.fixture(params=[False, True], ids=["offsets32", "addresses64"])
def rows64(request):
    """row maps through 32-bit buffer offsets (images below 4 GB) and, forced, through the 64-bit addresses of larger images"""
    lib = _lib.load()
    assert lib.gte_gemm_p3_set_rows64(1 if request.param else 0) == 0
    yield request.param
    assert lib.gte_gemm_p3_set_rows64(0) == 0


@pytest.mark.parametrize("rows,k,n", [(24437, 831, 512), (3000, 831, 512), (100, 363, 256), (1, 48, 256), (6001, 831, 256)])
@pytest.mark.parametrize("cfg", [None] + ALL_NT_CFGS)
def test_nt_through_a_row_map_is_bitwise_the_gathered_gemm(force_nt_cfg, rows64, rows, k, n, cfg):
    if cfg is not None:
        if rows != 3000 or rows64:
            pytest.skip("forced tile shapes: one problem size, 32-bit offsets (the 64-bit path has its own two tiles)")
    force_nt_cfg(cfg)
    n_res = 40000
    g = torch.Generator(device=DEV).manual_seed(rows)
    res = _resident_image(torch.randn(n_res, k, device=DEV, generator=g))
    w = ops.p3_from_f32(torch.randn(n, k, device=DEV, generator=g))
    bias = torch.randn(n, device=DEV, generator=g)
    rm = _row_map(n_res, rows, rows)
    mapped = ops.P3(res.data, rows, k, row_map=rm, res_rows=n_res)
    want = ops.gemm_p3_nt(mapped.gathered(), w, bias=bias, bias_cols=n // 2)
    got = ops.gemm_p3_nt(mapped, w, bias=bias, bias_cols=n // 2)
    assert torch.equal(got, want)


@pytest.mark.parametrize("rows,k,m", [(24437, 831, 256), (3000, 831, 256), (100, 363, 128), (17, 48, 256), (6001, 831, 128)])
def test_tn_through_a_row_map_is_bitwise_the_gathered_gemm(rows64, rows, k, m):
    """dW0 = [dz^T X | q^T X] with X = mapped rows of the resident image; the rows past the map's end read as zeros."""
    n_res = 40000
    g = torch.Generator(device=DEV).manual_seed(rows + 1)
    res = _resident_image(torch.randn(n_res, k, device=DEV, generator=g))
    dz, q = (ops.p3_from_f32(torch.randn(rows, m, device=DEV, generator=g)) for _ in range(2))
    rm = _row_map(n_res, rows, rows + 1)
    mapped = ops.P3(res.data, rows, k, row_map=rm, res_rows=n_res)
    want = ops.gemm_p3_tn(dz, mapped.gathered(), a2=q, two_segments=True)
    got = ops.gemm_p3_tn(dz, mapped, a2=q, two_segments=True)
    assert torch.equal(got, want)
    # last resident row in the map, one segment
    rm2 = rm.clone()
    rm2[0] = n_res - 1
    mapped2 = ops.P3(res.data, rows, k, row_map=rm2, res_rows=n_res)
    assert torch.equal(ops.gemm_p3_tn(dz, mapped2), ops.gemm_p3_tn(dz, mapped2.gathered()))


@pytest.mark.parametrize("rows,k,n", [(24437, 831, 256), (3000, 831, 96), (3000, 363, 160), (100, 313, 1000), (1, 48, 256), (6001, 781, 112)])
@pytest.mark.parametrize("cfg", [None] + ALL_NT_CFGS)
def test_nt_with_two_resident_images_behind_one_row_map(force_nt_cfg, rows64, rows, k, n, cfg):
    """z = [x | ahn][rows] W^T + b (gte_gemm_p3_nt_rows2: the input layer on its features and their cached mean aggregate) is bit for
    bit the two-segment product on the gathered rows; every tile configuration."""
    if cfg is not None:
        if rows != 3000 or k != 831 or rows64:
            pytest.skip("forced tile shapes: one problem size, 32-bit offsets (the 64-bit path has its own two tiles)")
    force_nt_cfg(cfg)
    n_res = 30000
    g = torch.Generator(device=DEV).manual_seed(rows + 7)
    res1 = _resident_image(torch.randn(n_res, k, device=DEV, generator=g))
    res2 = _resident_image(torch.randn(n_res, k, device=DEV, generator=g))
    kp = -(-k // 16) * 16
    wb = torch.zeros(n, 2 * kp, device=DEV)
    wb[:, :k], wb[:, kp:kp + k] = torch.randn(n, k, device=DEV, generator=g), torch.randn(n, k, device=DEV, generator=g)
    w = ops.P3(ops.p3_from_f32(wb).data, n, 2 * kp)
    bias = torch.randn(n, device=DEV, generator=g)
    rm = _row_map(n_res, rows, rows)
    m1 = ops.P3(res1.data, rows, k, row_map=rm, res_rows=n_res)
    m2 = ops.P3(res2.data, rows, k, row_map=rm, res_rows=n_res)
    want = ops.gemm_p3_nt(m1.gathered(), w, a2=m2.gathered(), bias=bias)
    got = ops.gemm_p3_nt(m1, w, a2=m2, bias=bias)
    assert torch.equal(got, want)


@pytest.mark.parametrize("rows,k,m", [(24437, 831, 256), (3000, 831, 96), (100, 363, 160), (17, 48, 256), (6001, 781, 100)])
def test_tn_with_two_resident_images_behind_one_row_map(rows64, rows, k, m):
    """dW0 = [dz^T x | dz^T ahn] with x / ahn = mapped rows of two resident images (gte_gemm_p3_tn_rows2), 32-bit offsets and 64-bit
    addresses: bit for bit the product on the gathered rows."""
    n_res = 30000
    g = torch.Generator(device=DEV).manual_seed(rows + 3)
    res1 = _resident_image(torch.randn(n_res, k, device=DEV, generator=g))
    res2 = _resident_image(torch.randn(n_res, k, device=DEV, generator=g))
    dz = ops.p3_from_f32(torch.randn(rows, m, device=DEV, generator=g))
    rm = _row_map(n_res, rows, rows + 1)
    m1 = ops.P3(res1.data, rows, k, row_map=rm, res_rows=n_res)
    m2 = ops.P3(res2.data, rows, k, row_map=rm, res_rows=n_res)
    want = ops.gemm_p3_tn(dz, m1.gathered(), b2=m2.gathered(), two_segments=True)
    got = ops.gemm_p3_tn(dz, m1, b2=m2, two_segments=True)
    assert torch.equal(got, want)


@pytest.mark.parametrize("m,n,k", [(24437, 256, 831), (3000, 218, 13), (3000, 96, 363), (700, 100, 781), (100, 139, 63), (1, 5, 48),
                                   (260, 256, 16)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("mapped", [False, True], ids=["dense", "rows2"])
def test_nt_with_layernorm_forward_epilogue_is_bitwise_the_two_launches(rows64, ln_rows, wlayout, m, n, k, relu, mapped):
    """gte_gemm_p3_nt_ln_fwd / gte_gemm_p3_nt_rows2_ln_fwd: z = [a1 | a2] b^T + bias, the row statistics, y as fp32 and as a P3 image
    are bit for bit what gte_gemm_p3_nt (+ _rows2) followed by gte_ln_relu_fwd_p3 write; padding columns zero."""
    if rows64 and not mapped:
        pytest.skip("64-bit addresses: row-mapped operands only")
    lib, P = _lib.load(), _lib.ptr
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    n_res = max(2 * m, 64)
    x1, x2 = torch.randn(n_res, k, device=DEV, generator=g), torch.randn(n_res, k, device=DEV, generator=g)
    kp, ld = -(-k // 16) * 16, -(-n // 16) * 16
    wb = torch.zeros(n, 2 * kp, device=DEV)
    wb[:, :k], wb[:, kp:kp + k] = torch.randn(n, k, device=DEV, generator=g) * 0.1, torch.randn(n, k, device=DEV, generator=g) * 0.1
    w = ops.P3(ops.p3_from_f32(wb).data, n, 2 * kp)
    bias, gamma, beta = (torch.randn(n, device=DEV, generator=g) for _ in range(3))
    if mapped:
        rm = _row_map(n_res, m, m + 5)
        a1 = ops.P3(_resident_image(x1).data, m, k, row_map=rm, res_rows=n_res)
        a2 = ops.P3(_resident_image(x2).data, m, k, row_map=rm, res_rows=n_res)
    else:
        a1, a2 = ops.p3_from_f32(x1[:m].contiguous()), ops.p3_from_f32(x2[:m].contiguous())

    def bufs():
        z = torch.zeros((m, ld), dtype=torch.float32, device=DEV)
        y = torch.full((m, ld), 7.0, dtype=torch.float32, device=DEV)
        y[:, n:] = 0
        yp = ops.P3.empty(m, n, DEV)
        yp.data.fill_(0x55)
        return z, y, yp, torch.zeros(2 * m, dtype=torch.float32, device=DEV)
    z0, y0, yp0, st0 = bufs()
    ops.gemm_p3_nt(a1, w, a2=a2, bias=bias, out=z0[:, :n])
    _lib.check(lib.gte_ln_relu_fwd_p3(P(z0), ld, P(gamma), P(beta), 1e-5, int(relu), P(y0), ld, P(yp0.data), yp0.ldp, P(st0), m, n,
                                      _lib.current_stream()), "ln_relu_fwd_p3")
    z1, y1, yp1, st1 = bufs()
    wl = _weights(w, wlayout)              # (block-major: the block-major-weights kernel where it applies, gemm_p3_nt_sq_kernel)
    ops.gemm_p3_nt_ln_fwd(a1, wl, a2, bias, gamma, beta, 1e-5, relu, z1, y=y1, yp3=yp1, stats=st1)
    assert torch.equal(z1, z0) and torch.equal(st1, st0) and torch.equal(y1, y0) and torch.equal(yp1.data, yp0.data)
    # image only (what the step asks for when the next layer takes an image)
    z2, _, yp2, st2 = bufs()
    ops.gemm_p3_nt_ln_fwd(a1, wl, a2, bias, gamma, beta, 1e-5, relu, z2, y=None, yp3=yp2, stats=st2)
    assert torch.equal(z2, z0) and torch.equal(yp2.data, yp0.data)


def test_row_map_entry_points_validate():
    lib = _lib.load()
    z = torch.zeros(64, 96, dtype=torch.uint8, device=DEV)
    c = torch.zeros(16, 16, device=DEV)
    P = _lib.ptr
    assert lib.gte_gemm_p3_nt_rows(P(z), 96, 16, None, 64, P(z), 96, None, 0, P(c), 16, 16, 16, 0, 0, None) == -1
    rm = torch.zeros(64, dtype=torch.int32, device=DEV)
    assert lib.gte_gemm_p3_nt_rows(P(z), 96, 16, P(rm), 0, P(z), 96, None, 0, P(c), 16, 16, 16, 0, 0, None) == -1
    assert b"empty resident image" in lib.gte_last_error()
    assert lib.gte_gemm_p3_set_rows64(2) == -1


def test_row_maps_into_a_resident_image_above_4_gb():
    """900 000 resident rows x 831 features = 4.5 GB of image: 32-bit buffer offsets do not reach its rows, the GEMMs follow the
    map through 64-bit per-lane addresses (global_load_lds).  Forward (NT) and weight-gradient (TN) products against the same
    GEMMs on a gathered copy of the rows, bit for bit; the map takes rows from the whole image, its last row included."""
    n_res, k, rows = 900_000, 831, 24437
    ldp = _lib.load().gte_p3_row_bytes(k)
    assert n_res * ldp > (1 << 32)
    img = ops.P3.empty(n_res, k, DEV, rows_cap=n_res + 1)
    img.data[n_res:].zero_()
    g = torch.Generator(device=DEV).manual_seed(5)
    for r0 in range(0, n_res, 100_000):                            # (converted chunk by chunk: 330 MB of fp32 at a time)
        chunk = torch.randn(100_000, k, device=DEV, generator=g)
        ops.p3_from_f32(chunk, out=img, row0=r0)
        del chunk
    rm = _row_map(n_res, rows, 77)
    rm[0], rm[1] = n_res - 1, 0                                    # first and last row of the image
    assert int(rm[:rows].max()) * ldp > (1 << 32)
    mapped = ops.P3(img.data, rows, k, row_map=rm, res_rows=n_res)
    gathered = mapped.gathered()
    w = ops.p3_from_f32(torch.randn(512, k, device=DEV, generator=g))
    bias = torch.randn(512, device=DEV, generator=g)
    assert torch.equal(ops.gemm_p3_nt(mapped, w, bias=bias, bias_cols=256), ops.gemm_p3_nt(gathered, w, bias=bias, bias_cols=256))
    small = ops.P3(img.data, 3000, k, row_map=torch.cat([rm[:3000], rm[-33:]]), res_rows=n_res)      # the 128-row tile
    assert torch.equal(ops.gemm_p3_nt(small, w), ops.gemm_p3_nt(small.gathered(), w))
    dz, q = (ops.p3_from_f32(torch.randn(rows, 256, device=DEV, generator=g)) for _ in range(2))
    assert torch.equal(ops.gemm_p3_tn(dz, mapped, a2=q, two_segments=True), ops.gemm_p3_tn(dz, gathered, a2=q, two_segments=True))


# ---- the backward GEMM with the LayerNorm backward of the layer below as its epilogue ---------------------------------------
@pytest.mark.parametrize("m,n,k,relu", [(24437, 256, 256, True), (3000, 256, 256, False), (129, 128, 64, True), (1, 256, 256, True),
                                        (40000, 256, 128, True), (500, 144, 256, True), (40000, 256, 256, True), (80000, 256, 256, True),
                                        (80149, 96, 96, False), (3000, 256, 304, True), (24437, 160, 1008, False)])
def test_nt_with_layernorm_backward_epilogue_is_bitwise_the_two_launches(ln_rows, wlayout, m, n, k, relu):
    """gte_gemm_p3_nt_ln_bwd: dy = [dz1 | q1] [W_s^T | W_n^T]^T is never stored; dz0 (fp32 and image) must be bit for bit what
    gte_gemm_p3_nt + gte_ln_relu_bwd_p3 produce, the column sums agree to summation order.  (k = 304: 19 + 19 K blocks -- with
    block-major weights a 64-deep slot of the block-major-weights kernel straddles the two K segments and the last slot is short;
    k = 1008: 126 blocks, the H = 1000 shapes' depth.)"""
    g = torch.Generator(device=DEV).manual_seed(m + n)
    rnd = lambda *sh: torch.randn(*sh, device=DEV, generator=g)
    a1, a2, w = rnd(m, k), rnd(m, k), rnd(n, 2 * k) / 16
    z = rnd(m, 2 * n)[:, :n]                                 # a strided view: z lives in the left half of t = [t_self | t_neigh]
    gam, bet = 1 + 0.1 * rnd(n), 0.1 * rnd(n)
    mu = z.mean(1)
    stats = torch.cat([mu, 1.0 / torch.sqrt(z.var(1, unbiased=False) + 1e-5)]).contiguous()
    a1p, a2p, wp = ops.p3_from_f32(a1), ops.p3_from_f32(a2), ops.p3_from_f32(w)
    # two launches
    dy = ops.gemm_p3_nt(a1p, wp, a2=a2p)
    lib, P = _lib.load(), _lib.ptr
    dz_ref, dzp_ref = torch.empty(m, n, device=DEV), ops.P3.empty(m, n, DEV)
    dg_ref, db_ref, dbias_ref = (torch.zeros(n, device=DEV) for _ in range(3))
    ws = torch.empty(int(lib.gte_ln_relu_bwd_workspace_bytes(m, n)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.gte_ln_relu_bwd_p3(P(dy), n, P(z), z.stride(0), P(stats), P(gam), P(bet), int(relu), P(dz_ref), n, P(dzp_ref.data),
                                      dzp_ref.ldp, P(dg_ref), P(db_ref), P(dbias_ref), m, n, P(ws), ws.numel(), _lib.current_stream()),
               "gte_ln_relu_bwd_p3")
    # one launch
    dz, dzp = torch.full((m, n), 7.0, device=DEV), ops.P3.empty(m, n, DEV)
    dg, db, dbias = (torch.full((n,), 7.0, device=DEV) for _ in range(3))
    wp = _weights(wp, wlayout)
    ops.gemm_p3_nt_ln_bwd(a1p, wp, a2p, z, stats, gam, bet, relu, dz, dzp, dg, db, dbias)
    assert torch.equal(dz, dz_ref)
    # image only (an input layer on the cached aggregate: its dW reads the image, nothing reads the fp32 rows)
    dzp_only = ops.P3.empty(m, n, DEV)
    dzp_only.data.fill_(0x55)
    ops.gemm_p3_nt_ln_bwd(a1p, wp, a2p, z, stats, gam, bet, relu, None, dzp_only, *(torch.zeros(n, device=DEV) for _ in range(3)))
    assert torch.equal(dzp_only.data, dzp_ref.data)
    assert torch.equal(ops.p3_to_f32(dzp), dz_ref)
    for got, want in ((dg, dg_ref), (db, db_ref), (dbias, dbias_ref)):
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(want.abs().max()) + 1e-6)


@pytest.mark.parametrize("m,n,relu", [(24437, 218, True), (3000, 149, True), (5001, 206, False), (24437, 100, True), (777, 157, True),
                                      (260, 5, True), (2, 250, True)])
def test_nt_with_layernorm_backward_epilogue_at_widths_that_are_not_multiples_of_16(ln_rows, wlayout, m, n, relu):
    """The hidden widths the reference's runs use (218, 206, 157, 149, 100): rows of z / dz padded to 16 floats, the LayerNorm over
    the true width.  dz -- fp32 with its padding, and the image up to the next multiple of 16 columns -- bit for bit what
    gte_gemm_p3_nt + gte_ln_relu_bwd_p3 (the general-width kernel) produce; k = n as in the step (dX of the layer above)."""
    ld = -(-n // 16) * 16
    g = torch.Generator(device=DEV).manual_seed(m + n)
    rnd = lambda *sh: torch.randn(*sh, device=DEV, generator=g)
    a1, a2, w = rnd(m, n), rnd(m, n), rnd(n, 2 * ld) / 16
    w[:, n:ld] = 0                                             # [W_s^T | W_n^T] with the K blocks of each half padded to 16
    w[:, ld + n:] = 0
    t = torch.zeros(m, 2 * ld, device=DEV)
    t[:, :n] = rnd(m, n)
    z = t[:, :n]                                               # the left half of t = [t_self | t_neigh], padding zero
    gam, bet = 1 + 0.1 * rnd(n), 0.1 * rnd(n)
    mu = z.mean(1)
    stats = torch.cat([mu, 1.0 / torch.sqrt(z.var(1, unbiased=False) + 1e-5)]).contiguous()
    a1p, a2p = ops.p3_from_f32(a1), ops.p3_from_f32(a2)
    wp = ops.p3_from_f32(w)                                    # K = 2 ld: block ld / 16 starts the second segment
    lib, P = _lib.load(), _lib.ptr
    dyb = torch.zeros(m, ld, device=DEV)
    ops.gemm_p3_nt(a1p, wp, a2=a2p, out=dyb[:, :n])
    dz_ref, dzp_ref = torch.zeros(m, ld, device=DEV), ops.P3.empty(m, n, DEV)
    dzp_ref.data.zero_()
    dg_ref, db_ref, dbias_ref = (torch.zeros(n, device=DEV) for _ in range(3))
    ws = torch.empty(int(lib.gte_ln_relu_bwd_workspace_bytes(m, n)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.gte_ln_relu_bwd_p3(P(dyb), ld, P(t), 2 * ld, P(stats), P(gam), P(bet), int(relu), P(dz_ref), ld, P(dzp_ref.data),
                                      dzp_ref.ldp, P(dg_ref), P(db_ref), P(dbias_ref), m, n, P(ws), ws.numel(), _lib.current_stream()),
               "gte_ln_relu_bwd_p3")
    dzb, dzp = torch.zeros(m, ld, device=DEV), ops.P3.empty(m, n, DEV)
    dzp.data.fill_(0x55)                                       # the fused launch must write every column block of the image itself
    dg, db, dbias = (torch.full((n,), 7.0, device=DEV) for _ in range(3))
    ops.gemm_p3_nt_ln_bwd(a1p, _weights(wp, wlayout), a2p, z, stats, gam, bet, relu, dzb[:, :n], dzp, dg, db, dbias)
    assert torch.equal(dzb, dz_ref)
    assert torch.equal(dzp.data, dzp_ref.data)
    assert float(dzb[:, n:].abs().sum()) == 0.0
    for got, want in ((dg, dg_ref), (db, db_ref), (dbias, dbias_ref)):
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(want.abs().max()) + 1e-6)


@pytest.mark.parametrize("m,n,k,kin", [(24437, 256, 256, 13), (3000, 256, 256, 13), (129, 128, 64, 9), (1, 256, 256, 13), (40000, 256, 128, 14)])
def test_nt_with_the_short_input_layers_backward_as_epilogue(m, n, k, kin):
    """gte_gemm_p3_nt_smallk_bwd against gte_gemm_p3_nt + gte_sage_smallk_bwd (same per-row arithmetic, csrc/smallk_step.h; the
    partial sums are grouped by 128-row tiles instead of 64-row blocks: summation order only)."""
    g = torch.Generator(device=DEV).manual_seed(m + n)
    rnd = lambda *sh: torch.randn(*sh, device=DEV, generator=g)
    a1, a2, w = rnd(m, k), rnd(m, k), rnd(n, 2 * k) / 16
    x, ahn = rnd(m, kin), rnd(m, kin)
    W0, b0 = rnd(n, 2 * kin) / 5, rnd(n)
    gam, bet = 1 + 0.1 * rnd(n), 0.1 * rnd(n)
    lib, P = _lib.load(), _lib.ptr
    y, stats = torch.empty(m, n, device=DEV), torch.empty(2 * m, device=DEV)
    _lib.check(lib.gte_sage_linear_fwd(P(x), kin, kin, P(ahn), kin, kin, P(W0), 2 * kin, P(b0), P(gam), P(bet), 1e-5, 1, None, n,
                                       P(stats), P(y), n, m, n, _lib.current_stream()), "fwd")
    a1p, a2p, wp = ops.p3_from_f32(a1), ops.p3_from_f32(a2), ops.p3_from_f32(w)
    # two launches
    dy = ops.gemm_p3_nt(a1p, wp, a2=a2p)
    gW_ref = torch.empty(n, 2 * kin, device=DEV)
    gb_ref, gg_ref, gbe_ref = (torch.empty(n, device=DEV) for _ in range(3))
    ops.sage_smallk_bwd(dy, x, ahn, W0, b0, gam, bet, stats, True, gW_ref, gb_ref, gg_ref, gbe_ref)
    # one launch
    gW = torch.full((n, 2 * kin), 7.0, device=DEV)
    gb, gg, gbe = (torch.full((n,), 7.0, device=DEV) for _ in range(3))
    ops.gemm_p3_nt_smallk_bwd(a1p, wp, a2p, x, ahn, W0, b0, gam, bet, stats, True, gW, gb, gg, gbe)
    for got, want in ((gW, gW_ref), (gb, gb_ref), (gg, gg_ref), (gbe, gbe_ref)):
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(want.abs().max()) + 1e-6)


# ---- operands outside the comfortable range -------------------------------------------------------------------------------
def test_non_finite_operands_poison_exactly_the_outputs_fp32_poisons(split_mode):
    """inf cannot be cut into pieces (inf - inf is NaN) and 0 x inf appears among the six products, so an output that fp32
    arithmetic makes +-inf or NaN comes out NaN here -- never a finite number; every output whose operands are all finite is
    untouched (bit-identical to the GEMM without the poisoned rows)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    a, b = torch.randn(300, 100, device=DEV, generator=g), torch.randn(256, 100, device=DEV, generator=g)
    clean = ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b))
    a2 = a.clone()
    a2[7, 3], a2[100, 50], a2[200, 99] = float("inf"), float("-inf"), float("nan")
    got = ops.gemm_p3_nt(ops.p3_from_f32(a2), ops.p3_from_f32(b))
    f32 = a2 @ b.t()
    assert torch.equal(torch.isfinite(got), torch.isfinite(f32))
    assert torch.isnan(got[[7, 100, 200]]).all()
    keep = torch.ones(300, dtype=torch.bool, device=DEV)
    keep[[7, 100, 200]] = False
    assert torch.equal(got[keep], clean[keep])
    # the same through the weight-gradient GEMM: a poisoned k row poisons every output
    d = torch.randn(300, 128, device=DEV, generator=g)
    tn = ops.gemm_p3_tn(ops.p3_from_f32(d), ops.p3_from_f32(a2))
    assert not torch.isfinite(tn[:, [3, 50, 99]]).any() and torch.isfinite(tn[:, :3]).all()


@pytest.mark.parametrize("e", [-20, -60, -100, -110, -126, -140])
def test_small_magnitudes(split_mode, e):
    """Operands scaled by 2^e against fp64.  The pieces keep fp32's exponent range, so down to |x| ~ 2^-110 the result has full
    accuracy (units of u = 2^-24 sum |a_k b_k|); below that the low pieces (2^-16 |x|) fall under bf16's normal range 2^-126
    and are flushed: the error is bounded by K x 2^-126 x 2^-7 x max|b| ABSOLUTE, which the test states.  fp32 denormal inputs
    (e = -140) are flushed to zero by the conversion: the product is 0."""
    g = torch.Generator(device=DEV).manual_seed(-e)
    k = 256
    a = torch.randn(200, k, device=DEV, generator=g) * 2.0 ** max(e, -126)
    if e < -126:
        a = a * 2.0 ** (e + 126)
    b = torch.randn(128, k, device=DEV, generator=g)
    c = ops.gemm_p3_nt(ops.p3_from_f32(a), ops.p3_from_f32(b))
    ref = a.double() @ b.double().t()
    unit = 2.0 ** -24 * (a.double().abs() @ b.double().abs().t())
    err = (c.double() - ref).abs()
    if e >= -100:
        assert float((err / unit).max()) < 6.0
    else:
        # absolute: every piece below 2^-126 may be dropped: |x| 2^-8 at worst for the middle piece of an operand just above
        # the flush threshold, i.e. <= 2^-126 per term
        bound = k * 2.0 ** -126 * float(b.abs().max()) + 6.0 * unit
        assert bool((err <= bound).all())
    assert torch.isfinite(c).all()
