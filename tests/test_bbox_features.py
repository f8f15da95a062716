"""BBOX node features: oracle hand cases on CPU; HIP kernel bit-exact vs the oracle on GPU."""
import numpy as np
import pytest

from oracle import bbox_features as ob


def test_oracle_hand_cases():
    assert ob.char_counts("Table 3.1") == (5, 2, 1)
    assert ob.char_counts("   ") == (0, 0, 0)
    f = ob.bbox_features([[10, 20, 45, 31]], [ob.char_counts("ab1")])[0]
    # w=35 h=11 cx=45-int(17.5)=28 cy=31-int(5.5)=26 area=385
    np.testing.assert_array_equal(f[:9], np.float32([35, 11, 28, 26, 385, 10, 20, 45, 31]))
    h = ob.histogram((2, 1, 0))
    assert sum(h) == 1.0 and h[3] == 0.0 and abs(h[0] - 2 / 3) < 1e-15
    assert ob.histogram((0, 0, 0)) == [0.0, 0.0, 0.0, 1.0]
    assert ob.histogram((1, 1, 1))[:3] != [1 / 3, 1 / 3, 1 / 3] or sum(ob.histogram((1, 1, 1))) == 1.0
    # negative width (degenerate box): int() truncates toward zero, as Python does
    g = ob.bbox_features([[50, 0, 45, 9]], [(0, 0, 0)])[0]
    assert g[0] == -5 and g[2] == 45 - int(-5 / 2) == 47


@pytest.mark.gpu
def test_kernel_is_bit_exact():
    import torch
    from gnn_tableextraction_amd import _lib
    rng = np.random.default_rng(0)
    n = 50_000
    x0, y0 = rng.integers(0, 1600, n), rng.integers(0, 2300, n)
    bbox = np.stack([x0, y0, x0 + rng.integers(-3, 400, n), y0 + rng.integers(0, 60, n)], 1).astype(np.int32)
    counts = rng.integers(0, 12, (n, 3)).astype(np.int32)
    counts[:500] = 0                                            # empty words
    counts[500:1500, 1:] = 0                                    # letters only
    want = ob.bbox_features(bbox, counts)
    lib = _lib.load()
    b, c = torch.from_numpy(bbox).cuda(), torch.from_numpy(counts).cuda()
    out = torch.zeros(n, 16, device="cuda")
    _lib.check(lib.gte_bbox_features(_lib.ptr(b), _lib.ptr(c), _lib.ptr(out), 16, n, _lib.current_stream()))
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, :13], want)            # bit-exact
    assert (got[:, 13:] == 0).all()
    assert (np.float64(got[:, 9:13]).sum(1) - 1.0).__abs__().max() < 1e-6
