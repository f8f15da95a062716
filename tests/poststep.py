"""Post-step comparison shared by the CPU and GPU golden tests.

The first Adam step maps a (weight-decayed) gradient g to  lr * g / (|g| + eps)  with eps = 1e-8: wherever |g| >> eps the
update is +-lr whatever the last bits of g are, but entries whose true gradient is ~0 get a gradient of +-1e-9 from
summation-order noise alone, and their update lands anywhere in [-lr, lr] -- between two runs of the REFERENCE on different
thread counts too.  So:
  * every parameter must equal the reference's post-step value within ``step_tolerance``: 1e-5 wherever its decayed gradient
    is resolved against the summation noise, opening up continuously (to at most 2 lr) only where it is not; the entries
    beyond 1e-5 must stay below 2 % of each tensor (at full size ~0.4 % of layer 0's weights have a gradient that cancels
    against the weight-decay term to < 5e-7);
  * the logits after the step are compared at 1e-4 against the reference forward (the CPU oracle, itself pinned to the
    reference's forward at 1e-5) run on the reference's post-step state with exactly those ill-conditioned entries taken
    from the state under test -- i.e. everything except the entries that are noise in the reference itself is checked tight."""
import numpy as np
import torch

from oracle import gcnsage_cpu as oc


NOISE = 2.0 ** -18        # bound on the summation-order noise of a gradient entry, relative to the tensor's largest entry
                          # (measured between the reference's own runs: ~2^-21; fp32 sums over 10^3 - 10^4 nodes)


def step_tolerance(g_eff, g_absmax, lr=0.01, eps=1e-8):
    """Allowed |difference| of a parameter after ONE Adam step, per entry.  The step maps the decayed gradient g to
    u(g) = lr g / (|g| + eps); a perturbation dg of the gradient moves it by  u'(g) dg = lr eps / (|g| + eps)^2 dg.  With dg
    bounded by NOISE x the tensor's largest gradient entry this is 1e-5 for every resolved gradient and opens up -- continuously,
    up to 2 lr -- only where |g| is itself of the size of the noise.  A wrong-sign update at |g| = 1e-6 (error ~ 2 lr) is ~1000x
    outside it (the flat |g| < 1e-5 mask of round 2 let it through)."""
    g = np.abs(np.asarray(g_eff, dtype=np.float64))
    sens = lr * eps / (g + eps) ** 2
    return 1e-5 + np.minimum(2 * lr * (1 + 1e-4), sens * NOISE * float(g_absmax))


def hybrid_state(want: dict, got: dict, g_eff: dict, lr=0.01, g_absmax: dict = None):
    """want/got/g_eff: name -> ndarray (reference post-step params, ours, decayed gradient).  Entries that differ by more than
    1e-5 must be inside ``step_tolerance`` and stay below 2 % of their tensor; they are taken from the state under test."""
    hyb = {}
    for k, w in want.items():
        o = np.asarray(got[k])
        amax = float(np.abs(g_eff[k]).max()) if g_absmax is None else float(g_absmax[k])
        diff = np.abs(o.astype(np.float64) - w)
        assert (diff <= step_tolerance(g_eff[k], amax, lr) + 1e-5 * np.abs(w)).all(), \
            f"{k}: a parameter differs after the step by more than its gradient's conditioning allows (max {diff.max():.3e})"
        bad = ~np.isclose(o, w, rtol=1e-5, atol=1e-5)
        assert bad.sum() <= 2e-2 * bad.size + 2, f"{k}: {int(bad.sum())} of {bad.size} entries are ill-conditioned"
        h = w.copy()
        h[bad] = o[bad]
        hyb[k] = torch.from_numpy(h)
    return hyb


def check_against_fixture(z, params_after: dict, logits_after: np.ndarray, wd=5e-4, lr=0.01, atol=1e-4):
    """``z``: a golden case of oracle/make_golden.py; params_after: state_dict-style name -> ndarray after ONE step."""
    want = {k[len("state1."):]: z[k] for k in z.files if k.startswith("state1.")}
    g_eff = {k: (z["grad." + k] + wd * z["state0." + k]) if ("grad." + k) in z.files else np.zeros_like(v)
             for k, v in want.items()}
    hyb = hybrid_state(want, params_after, g_eff, lr)
    n = int(z["meta"][0])
    og = oc.OracleGraph(z["src"], z["dst"], n, z["w"])
    ref_after = oc.gcnsage_forward(hyb, og, torch.from_numpy(z["x"])).numpy()
    err = float(np.abs(np.asarray(logits_after) - ref_after).max())
    assert err < atol, f"post-step logits differ by {err}"
    # and the fixture's own post-step logits (the reference's noise entries included) stay within the loose bound
    assert np.abs(np.asarray(logits_after) - z["logits_after_step"]).max() < 5e-3
    return err


# ---- the trimmed headline-shape fixture (oracle/make_golden.py: run_headline_case) -----------------------------------------
def headline_case(golden_dir):
    """(z, src, dst, w, x, y, state0): inputs regenerated from the seed, initial weights through the model constructor
    (host code; the repository reproduces the reference's RNG stream -- the fixture's per-tensor sums check it)."""
    return trimmed_case(golden_dir, "headline_n2000_f831_h256")


def trimmed_case(golden_dir, name):
    """A trimmed fixture of oracle/make_golden.py (the headline shape, or one of the reference's run shapes ``shape_*``):
    (z, src, dst, w, x, y, state0, model)."""
    import os
    import gnn_tableextraction_amd as gte
    from oracle.make_golden import headline_inputs, shape_inputs
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = [int(v) for v in z["meta"]]
    n, f0, hid, ncls, nl, seed = meta[:6]
    if name.startswith("shape_"):
        src, dst, w, x, y = shape_inputs(seed, n, f0, bool(meta[6]))
    else:
        src, dst, w, x, y = headline_inputs(seed, n, f0)
    assert np.array_equal(src, z["src"]) and np.array_equal(dst, z["dst"]) and np.array_equal(w, z["w"])
    assert float(x.astype(np.float64).sum()) == float(z["x_sum"]) and int(y.sum()) == int(z["y_sum"])
    torch.manual_seed(seed)
    model = gte.GcnSAGE(f0, hid, ncls, nl, torch.nn.functional.relu, 0)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in state0.items():
        # (the generator's own summation: numpy's pairwise fp64 sum -- torch's order differs in the last bits on 2 M elements)
        assert float(v.numpy().astype(np.float64).sum()) == float(z["state0_sum." + k]), f"{k}: not the reference's initial weights"
    return z, src, dst, w, x, y, state0, model


def check_headline(z, logits, hidden, loss, grads: dict, params_after: dict, logits_after, state0: dict, oracle_after=None):
    """Everything the trimmed fixture holds.  ``hidden``: per-layer activations [n, width]; grads / params_after: name -> ndarray."""
    np.testing.assert_allclose(logits, z["logits"], rtol=1e-5, atol=1e-5)                 # north_star tolerance
    for i, h in enumerate(hidden):
        np.testing.assert_allclose(np.asarray(h)[::16], z[f"hidden_rows16.{i}"], rtol=1e-5, atol=2e-5)
    assert abs(float(loss) - float(z["loss"])) < 1e-5
    for k, g in grads.items():
        g = np.asarray(g)
        if ("grad." + k) in z.files:
            ref = z["grad." + k]
            np.testing.assert_allclose(g, ref, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(ref).max())
        else:
            ref, idx = z["grad_val." + k], z["grad_idx." + k]
            amax = float(z["grad_absmax." + k])
            np.testing.assert_allclose(g.reshape(-1)[idx], ref, rtol=1e-4, atol=1e-6 + 1e-4 * amax)
            assert abs(float(g.astype(np.float64).sum()) - float(z["grad_sum." + k])) < 1e-4 * amax * np.sqrt(g.size)
            assert abs(float(np.abs(g).max()) - amax) <= 1e-4 * amax
    for k, p in params_after.items():
        p = np.asarray(p)
        if ("state1." + k) in z.files:
            want, got, p0 = z["state1." + k], p, state0[k].numpy()
            ge = (z["grad." + k] + 5e-4 * p0) if ("grad." + k) in z.files else np.zeros_like(want)
            amax = float(np.abs(z["grad." + k]).max()) if ("grad." + k) in z.files else 0.0
        else:
            idx = z["state1_idx." + k]
            want, got = z["state1_val." + k], p.reshape(-1)[idx]
            # the decayed gradient at the sampled state entries, from OUR gradient (checked against the reference above)
            ge = np.asarray(grads[k]).reshape(-1)[idx] + 5e-4 * state0[k].numpy().reshape(-1)[idx]
            amax = float(z["grad_absmax." + k])
        diff = np.abs(got.astype(np.float64) - want)
        assert (diff <= step_tolerance(ge, amax) + 1e-5 * np.abs(want)).all(), f"{k}: post-step parameter off by {diff.max():.3e}"
        bad = ~np.isclose(got, want, rtol=1e-5, atol=1e-5)
        assert bad.sum() <= 2e-2 * bad.size + 2, f"{k}: {int(bad.sum())} of {bad.size} entries are ill-conditioned"
    assert np.abs(np.asarray(logits_after) - z["logits_after_step"]).max() < 5e-3
    if oracle_after is not None:          # forward of the reference restatement on OUR post-step state: forward parity after the step
        assert np.abs(np.asarray(logits_after) - oracle_after).max() < 1e-4
