"""Thin Python wrappers + torch.autograd.Functions over the C ABI (include/gte.h).

PyTorch is plumbing here: it owns device memory and streams; every arithmetic step of the
hot path runs in libgte_hip.so.  Nothing in this file has a CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, current_stream, ptr, require_device

_ws_cache = {}

# ---- optional per-kernel timing (bench.py): HIP events on the launch stream around tagged calls ----
_timers = None          # None = off; dict tag -> {"events": [(start, end)], "work": float}


def enable_kernel_timers(on: bool = True) -> None:
    global _timers
    _timers = {} if on else None


TIMER_REPS = 5          # launches per timed region when the timers are on (see _timed)


def kernel_timer_report():
    """tag -> (launches, total_ms, total_work) ; synchronises."""
    out = {}
    if _timers:
        torch.cuda.synchronize()
        for tag, rec in _timers.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in rec["events"])
            out[tag] = (sum(r for _, _, r in rec["events"]), ms, rec["work"])
    return out


class _timed:
    """HIP-event pair around a kernel launch on the launch stream (off unless enable_kernel_timers).  A region whose
    body is written ``for _ in range(t.reps): launch()`` repeats its (idempotent) launch TIMER_REPS times inside one
    event pair when the timers are on: the ~5-10 us an event pair adds to a single 60 us kernel is amortised and the
    average is per launch."""
    __slots__ = ("tag", "work", "ev", "reps")

    def __init__(self, tag, work):
        self.tag, self.work = tag, work
        self.reps = 1

    def __enter__(self):
        if _timers is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()          # torch's current stream == the stream the kernel is launched on
        return self

    def repeat(self):
        """range() for an idempotent launch: TIMER_REPS launches when timing, one otherwise"""
        self.reps = TIMER_REPS if _timers is not None else 1
        return range(self.reps)

    def __exit__(self, *exc):
        if _timers is not None:
            self.ev[1].record()
            rec = _timers.setdefault(self.tag, {"events": [], "work": 0.0})
            rec["events"].append((self.ev[0], self.ev[1], self.reps))
            rec["work"] += self.work * self.reps
        return False


def _workspace(nbytes: int, device, tag: str = "ws") -> torch.Tensor:
    """Grow-only per-(device, stream, tag) scratch buffer (caller-provided workspace of the ABI)."""
    key = (device.index, current_stream(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _row_major(t: torch.Tensor) -> torch.Tensor:
    """2-D tensor whose rows are contiguous (arbitrary row stride is fine for the ABI)."""
    if t.dim() != 2:
        raise ValueError("expected a 2-D tensor")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


# --------------------------------------------------------------------------------------------------
# graph preparation
# --------------------------------------------------------------------------------------------------
def coo_to_csr(key: torch.Tensor, other: torch.Tensor, n: int, eweight: Optional[torch.Tensor] = None,
               row_scale: Optional[torch.Tensor] = None):
    """Stable sort of the COO by ``key`` on the device -> (indptr, indices, perm[, wout])."""
    require_device(key, "coo_to_csr")
    lib = _lib.load()
    e = key.numel()
    dev = key.device
    key = key.to(torch.int32).contiguous()
    other = other.to(torch.int32).contiguous()
    indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    indices = torch.empty(e, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    wout = torch.empty(e, dtype=torch.float32, device=dev) if (eweight is not None or row_scale is not None) else None
    nbytes = lib.gte_coo_to_csr_workspace_bytes(n, e)
    ws = _workspace(nbytes, dev, "csr")
    check(lib.gte_coo_to_csr(ptr(key), ptr(other), ptr(eweight), ptr(row_scale), n, e, ptr(indptr), ptr(indices),
                             ptr(perm), ptr(wout), ptr(ws), ws.numel(), current_stream()), "gte_coo_to_csr")
    return (indptr, indices, perm) if wout is None else (indptr, indices, perm, wout)


def inv_degree(indptr: torch.Tensor) -> torch.Tensor:
    require_device(indptr, "inv_degree")
    n = indptr.numel() - 1
    out = torch.empty(n, dtype=torch.float32, device=indptr.device)
    check(_lib.load().gte_inv_degree(ptr(indptr), ptr(out), n, current_stream()), "gte_inv_degree")
    return out


# --------------------------------------------------------------------------------------------------
# raw kernels
# --------------------------------------------------------------------------------------------------
class TilePlan:
    """Per-tile distinct-source lists of a CSR (graph structure; see gte_spmm_csr_tiled)."""
    __slots__ = ("tile_ptr", "tile_src", "local_index", "tile_rows", "max_unique")

    def __init__(self, tile_ptr, tile_src, local_index, tile_rows, max_unique):
        self.tile_ptr, self.tile_src, self.local_index = tile_ptr, tile_src, local_index
        self.tile_rows, self.max_unique = tile_rows, max_unique


def build_tile_plan(indptr: torch.Tensor, indices: torch.Tensor, n_rows: int) -> TilePlan:
    """Distinct sources per tile of ``gte_spmm_tile_rows()`` destination rows.  Index bookkeeping on the
    device (sort/unique through torch); done once per graph, reused by every layer / pass / epoch."""
    R = _lib.load().gte_spmm_tile_rows()
    dev = indptr.device
    nnz = indices.numel()
    ntiles = (n_rows + R - 1) // R
    if nnz == 0:
        z = torch.zeros(ntiles + 1, dtype=torch.int32, device=dev)
        return TilePlan(z, torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int16, device=dev), R, 0)
    deg = (indptr[1:] - indptr[:-1]).long()
    row_of_edge = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg)
    tile_of_edge = row_of_edge // R
    span = int(indices.max().item()) + 1
    key = tile_of_edge * span + indices.long()
    uniq, inv = torch.unique(key, return_inverse=True)                       # sorted by (tile, source)
    tile_of_u = uniq // span
    tile_ptr = torch.searchsorted(tile_of_u, torch.arange(ntiles + 1, device=dev)).to(torch.int32)
    local = inv - tile_ptr.long()[tile_of_edge]
    max_unique = int((tile_ptr[1:] - tile_ptr[:-1]).max().item())
    return TilePlan(tile_ptr.contiguous(), (uniq % span).to(torch.int32).contiguous(),
                    local.clamp(max=32767).to(torch.int16).contiguous(), R, max_unique)


TILED_MIN_FEATS = 32      # narrower rows (9, 13 features) leave most of a 16-lane row group idle: plain kernel
# The LDS-staged kernels win once the gathered matrix no longer lives in L2 (measured on MI355X, profiles/debug/
# tiled_crossover.py, page graphs of 200 ... 1600 pages):
#   widths that are a multiple of 32 with >= 4 chunks (spmm_tiled_full_kernel, two chunks in flight): ahead from ~50 MB
#     (48 k x 256: 20.8 vs 24.5 us; 191 k x 256: 88 vs 131 us; 389 k x 512: 316 vs 487 us; a 100-page batch, 25 MB: 15.3 vs 13.7);
#     the threshold stays at the size from which graphs carry tile plans at all (PageGraph.TILES_MIN_NODES);
#   other widths (spmm_tiled_kernel, one chunk in flight): behind the plain kernel on page graphs (in-degree ~6) at every size
#     (389 k x 831: 1042 vs 874 us), ahead on high-degree graphs (cfg4, in-degree 12: 1.05 vs 1.81 ms) -> needs TILED_MIN_DEGREE;
#   with the LayerNorm epilogue fused into the plain accumulating kernel the tiled route pays a separate LayerNorm pass:
#     it keeps the higher threshold.
TILED_MIN_BYTES = 192 << 20
TILED_FULL_MIN_BYTES = 96 << 20
TILED_MIN_DEGREE = 10


def use_tiled(n_rows: int, n_feat: int, nnz: Optional[int] = None, fused_ln: bool = False) -> bool:
    if n_feat < TILED_MIN_FEATS:
        return False
    nbytes = n_rows * n_feat * 4
    if n_feat % 32 == 0 and n_feat >= 128:
        return nbytes >= (TILED_MIN_BYTES if fused_ln else min(TILED_FULL_MIN_BYTES, TILED_MIN_BYTES))
    return nbytes >= TILED_MIN_BYTES and (nnz is None or nnz >= TILED_MIN_DEGREE * n_rows)


EDGE_PARALLEL_MIN_DEGREE = 64      # graphs whose largest row exceeds this take the edge-parallel kernel (gte_spmm_csr_edge)


def spmm_csr_edge(indptr, indices, weight, x: torch.Tensor, n_rows: int, mean: bool = False,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same contraction with the work split by EDGES (one wave per 64-edge segment of the CSR order, reduced segmented by
    destination row, two-pass carry for rows cut by a segment boundary): gte_spmm_csr_edge, for graphs with hub rows."""
    require_device(x, "spmm_csr_edge")
    lib = _lib.load()
    x = _row_major(x)
    if x.dtype != torch.float32:
        raise TypeError("spmm_csr_edge: fp32 features only")
    f, e = x.shape[1], int(indices.numel())
    if out is None:
        out = torch.empty((n_rows, f), dtype=torch.float32, device=x.device)
    ws = _workspace(lib.gte_spmm_csr_edge_workspace_bytes(e, f), x.device, "spmm_edge")
    nnz_bytes = 8.0 * e if weight is not None else 4.0 * e
    with _timed("spmm_edge", 2.0 * n_rows * f * 4 + nnz_bytes + 4.0 * (n_rows + 1)):
        check(lib.gte_spmm_csr_edge(ptr(indptr), ptr(indices), ptr(weight), ptr(x), _ld(x), ptr(out), _ld(out), n_rows, e, f,
                                    _lib.REDUCE_MEAN if mean else _lib.REDUCE_SUM, ptr(ws), ws.numel(), current_stream()),
              "gte_spmm_csr_edge")
    return out


def spmm_csr(indptr, indices, weight, x: torch.Tensor, n_rows: int, mean: bool = False,
             out: Optional[torch.Tensor] = None, accumulate: bool = False,
             tiles: Optional[TilePlan] = None, force_tiled: bool = False, max_degree: Optional[int] = None) -> torch.Tensor:
    """out[v] = scale_v * sum_e w[e] * x[indices[e]] over row v of the CSR (fp32 or bf16 features).
    With a ``TilePlan`` (and fp32 rows of >= 32 features) the LDS-staged kernel runs instead; with ``max_degree`` (the graph's
    largest row) above EDGE_PARALLEL_MIN_DEGREE the edge-parallel kernel (hub rows would serialise a lane group)."""
    require_device(x, "spmm_csr")
    lib = _lib.load()
    x = _row_major(x)
    if (max_degree is not None and max_degree > EDGE_PARALLEL_MIN_DEGREE and x.dtype == torch.float32 and not accumulate
            and not force_tiled):
        return spmm_csr_edge(indptr, indices, weight, x, n_rows, mean=mean, out=out)
    if tiles is not None and x.dtype == torch.float32 and (force_tiled or use_tiled(n_rows, x.shape[1], indices.numel())):
        f = x.shape[1]
        if out is None:
            if accumulate:
                raise ValueError("accumulate needs an output tensor")
            out = torch.empty((n_rows, f), dtype=x.dtype, device=x.device)
        nnz_bytes = 8.0 * indices.numel() if weight is not None else 4.0 * indices.numel()
        with _timed("spmm_tiled", 2.0 * n_rows * f * 4 + nnz_bytes + 4.0 * (n_rows + 1)):
            check(lib.gte_spmm_csr_tiled(ptr(indptr), ptr(indices), ptr(tiles.local_index), ptr(weight),
                                         ptr(tiles.tile_ptr), ptr(tiles.tile_src), ptr(x), _ld(x), ptr(out), _ld(out),
                                         n_rows, f, _lib.REDUCE_MEAN if mean else _lib.REDUCE_SUM, int(accumulate),
                                         current_stream()), "gte_spmm_csr_tiled")
        return out
    if x.dtype == torch.float32:
        dt = _lib.GTE_F32
    elif x.dtype == torch.bfloat16:
        dt = _lib.GTE_BF16
    else:
        raise TypeError(f"spmm_csr: unsupported dtype {x.dtype}")
    f = x.shape[1]
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an output tensor")
        out = torch.empty((n_rows, f), dtype=x.dtype, device=x.device)
    fn = lib.gte_spmm_csr_accumulate if accumulate else lib.gte_spmm_csr
    nnz_bytes = 8.0 * indices.numel() if weight is not None else 4.0 * indices.numel()
    with _timed("spmm_csr", 2.0 * n_rows * f * x.element_size() + nnz_bytes + 4.0 * (n_rows + 1)):
        check(fn(ptr(indptr), ptr(indices), ptr(weight), ptr(x), _ld(x), ptr(out), _ld(out), n_rows, f, dt,
                 _lib.REDUCE_MEAN if mean else _lib.REDUCE_SUM, current_stream()), "gte_spmm_csr")
    return out


def spmm_csr_p3(indptr, indices, weight, x: torch.Tensor, n_rows: int, mean: bool = False, out: Optional["P3"] = None) -> "P3":
    """The aggregation written straight as a P3 image (gte_spmm_csr_p3): out[v] = (1 / in_deg(v) if mean) sum_e w_e x[src(e)]."""
    require_device(x, "spmm_csr_p3")
    lib = _lib.load()
    x = _row_major(x)
    f = x.shape[1]
    if out is None:
        out = P3.empty(n_rows, f, x.device)
    check(lib.gte_spmm_csr_p3(ptr(indptr), ptr(indices), ptr(weight), ptr(x), _ld(x), ptr(out.data), out.ldp, n_rows, f,
                              _lib.REDUCE_MEAN if mean else _lib.REDUCE_SUM, current_stream()), "gte_spmm_csr_p3")
    return out


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
         out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """C = op(A) op(B) in fp32 on the MFMA path.  ``a``/``b`` are the STORED matrices."""
    require_device(a, "gemm")
    lib = _lib.load()
    a, b = _row_major(a), _row_major(b)
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    kb, n = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if k != kb:
        raise ValueError(f"gemm: inner dimensions differ ({k} vs {kb})")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    nbytes = lib.gte_gemm_workspace_bytes(m, n, k)
    ws = _workspace(nbytes, a.device, "gemm")
    with _timed("gemm_tn" if trans_a else "gemm_nn", 2.0 * m * n * k):
        check(lib.gte_gemm_f32(int(trans_a), int(trans_b), m, n, k, ptr(a), _ld(a), ptr(b), _ld(b), ptr(out),
                               _ld(out), int(accumulate), ptr(ws), ws.numel(), current_stream()), "gte_gemm_f32")
    return out


def sage_linear_dw(dz: torch.Tensor, x1: torch.Tensor, x2: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
    """out[n_out, k1+k2] = dz^T [x1 | x2] in one launch (weight gradient of the split-weight linear)."""
    lib = _lib.load()
    dz, x1 = _row_major(dz), _row_major(x1)
    m, n_out = dz.shape
    k1 = x1.shape[1]
    k2 = 0
    if x2 is not None:
        x2 = _row_major(x2)
        k2 = x2.shape[1]
    ws = _workspace(lib.gte_sage_linear_dw_workspace_bytes(n_out, k1, k2, m), dz.device, "gemm")
    with _timed("gemm_tn", 2.0 * m * n_out * (k1 + k2)):
        check(lib.gte_sage_linear_dw(ptr(dz), _ld(dz), ptr(x1), _ld(x1), k1, ptr(x2), 0 if x2 is None else _ld(x2), k2,
                                     ptr(out), _ld(out), n_out, m, ptr(ws), ws.numel(), current_stream()),
              "gte_sage_linear_dw")
    return out


def sage_linear_fwd(a1, a2, weight, bias, gamma, beta, eps: float, relu: bool, save_for_backward: bool):
    """y = relu?(LN?([a1|a2] W^T + b)); returns (y, z_save, stats)."""
    lib = _lib.load()
    a1 = _row_major(a1)
    m, k1 = a1.shape
    k2 = 0
    if a2 is not None:
        a2 = _row_major(a2)
        k2 = a2.shape[1]
    weight = _row_major(weight)
    n_out = weight.shape[0]
    dev = a1.device
    y = torch.empty((m, n_out), dtype=torch.float32, device=dev)
    ln = gamma is not None
    z = torch.empty((m, n_out), dtype=torch.float32, device=dev) if (ln and save_for_backward) else None
    stats = torch.empty(2 * m, dtype=torch.float32, device=dev) if (ln and save_for_backward) else None
    # two ABI calls (linear, then LayerNorm/ReLU) so each kernel can be timed on its own
    lin_out = z if z is not None else y
    with _timed("gemm_nt", 2.0 * m * (k1 + k2) * n_out):
        check(lib.gte_sage_linear_fwd(ptr(a1), _ld(a1), k1, ptr(a2), 0 if a2 is None else _ld(a2), k2,
                                      ptr(weight), _ld(weight), ptr(bias), None, None, float(eps),
                                      int(relu and not ln), None, 0, None, ptr(lin_out), _ld(lin_out), m, n_out,
                                      current_stream()), "gte_sage_linear_fwd")
    if ln:
        check(lib.gte_ln_relu_fwd(ptr(lin_out), _ld(lin_out), ptr(gamma), ptr(beta), float(eps), int(relu),
                                  ptr(y), _ld(y), ptr(stats), m, n_out, current_stream()), "gte_ln_relu_fwd")
    return y, z, stats


def ln_relu_bwd(dy, z, stats, gamma, beta, relu: bool, dgamma, dbeta, dbias) -> torch.Tensor:
    lib = _lib.load()
    dy = _row_major(dy)
    m, n = dy.shape
    dz = torch.empty((m, n), dtype=torch.float32, device=dy.device)
    nbytes = lib.gte_ln_relu_bwd_workspace_bytes(m, n)
    ws = _workspace(nbytes, dy.device, "lnbwd")
    check(lib.gte_ln_relu_bwd(ptr(dy), _ld(dy), ptr(z), 0 if z is None else _ld(z), ptr(stats), ptr(gamma), ptr(beta),
                              int(relu), ptr(dz), _ld(dz), ptr(dgamma), ptr(dbeta), ptr(dbias), m, n, ptr(ws),
                              ws.numel(), current_stream()), "gte_ln_relu_bwd")
    return dz


def sage_smallk_bwd(dy, a1, a2, weight, bias, gamma, beta, stats, relu: bool, dW, dbias=None, dgamma=None, dbeta=None):
    """LayerNorm(+ReLU) backward and dW = dz^T [a1 | a2] of the short-input INPUT layer in one pass (gte_sage_smallk_bwd): z is
    recomputed from the inputs, dz is never stored.  ``stats`` = the forward's [mean | rstd]."""
    lib = _lib.load()
    dy, a1, weight = _row_major(dy), _row_major(a1), _row_major(weight)
    m, n_out = dy.shape
    k1 = a1.shape[1]
    k2 = 0
    if a2 is not None:
        a2 = _row_major(a2)
        k2 = a2.shape[1]
    ws = _workspace(lib.gte_sage_smallk_bwd_workspace_bytes(m, k1 + k2, n_out), dy.device, "smallk_bwd")
    check(lib.gte_sage_smallk_bwd(ptr(dy), _ld(dy), ptr(a1), _ld(a1), k1, ptr(a2), 0 if a2 is None else _ld(a2), k2, ptr(weight),
                                  _ld(weight), ptr(bias), ptr(gamma), ptr(beta), ptr(stats), int(relu), ptr(dW), _ld(dW), ptr(dbias),
                                  ptr(dgamma), ptr(dbeta), m, n_out, ptr(ws), ws.numel(), current_stream()), "gte_sage_smallk_bwd")
    return dW


def weighted_ce(logits: torch.Tensor, labels: torch.Tensor, class_weight: Optional[torch.Tensor] = None,
                want_grad: bool = True, grad_scale: float = 1.0):
    """Returns (out3 = [loss, sum_w, n_correct] on the device, dlogits or None)."""
    require_device(logits, "weighted_ce")
    lib = _lib.load()
    logits = _row_major(logits)
    n, c = logits.shape
    if labels.dtype == torch.float32:
        lf = 1
    elif labels.dtype == torch.int64:
        lf = 0
    else:
        labels, lf = labels.to(torch.int64), 0
    labels = labels.contiguous()
    out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
    dl = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_grad else None
    ws = _workspace(lib.gte_weighted_ce_workspace_bytes(n), logits.device, "ce")
    check(lib.gte_weighted_ce(ptr(logits), _ld(logits), ptr(labels), lf, ptr(class_weight), n, c, float(grad_scale),
                              ptr(dl), 0 if dl is None else _ld(dl), ptr(out3), ptr(ws), ws.numel(), current_stream()),
          "gte_weighted_ce")
    return out3, dl


def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999,
              eps: float = 1e-8, weight_decay: float = 0.0, grad_scale: float = 1.0) -> None:
    require_device(param, "adam_step")
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError("adam_step: flat contiguous fp32 buffers required")
    check(_lib.load().gte_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), float(lr),
                                    float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                    float(grad_scale), current_stream()), "gte_adam_step")


# --------------------------------------------------------------------------------------------------
# autograd: aggregation  (models.py:53-54 + DGL GSpMM.backward)
# --------------------------------------------------------------------------------------------------
class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, graph, w, mean):
        csr = graph.in_csr()
        ctx.graph, ctx.w, ctx.mean = graph, w, mean
        return spmm_csr(csr.indptr, csr.indices, graph.in_weights(w), h, graph.num_nodes(), mean=mean,
                        tiles=graph.in_tiles(), max_degree=graph.max_in_degree())

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        rcsr = g.out_csr()
        # d h[u] = sum_{e: u->v} w_e * norm_v * dout[v]   (norm folded into the out-edge weights)
        dh = spmm_csr(rcsr.indptr, rcsr.indices, g.out_weights(ctx.w, ctx.mean), dout.contiguous(), g.num_nodes(),
                      tiles=g.out_tiles(), max_degree=g.max_out_degree())
        return dh, None, None, None


def aggregate(graph, h: torch.Tensor, w: Optional[torch.Tensor] = None, mean: bool = False) -> torch.Tensor:
    """``update_all(u_mul_e | copy_u, sum | mean)`` on a PageGraph, differentiable in ``h``."""
    require_device(h, "aggregate")
    return _Aggregate.apply(h, graph, w, mean)


# --------------------------------------------------------------------------------------------------
# autograd: one whole GcnSAGELayer  (models.py:46-72)
# --------------------------------------------------------------------------------------------------
class _SageLayer(torch.autograd.Function):
    """y = act(LN([h | (A_w h) * norm] W^T + b)) with every piece in HIP:
    aggregation -> split-weight MFMA GEMM (no concat) -> LayerNorm/ReLU; hand-written backward."""

    @staticmethod
    def forward(ctx, h, weight, bias, gamma, beta, graph, w, relu, eps, use_pp):
        h = _row_major(h)
        # grad mode is off inside Function.forward: ask the ctx which inputs need a gradient
        need_grad = any(ctx.needs_input_grad[:5])
        ahn = None
        if not use_pp:
            csr = graph.in_csr()
            ahn = spmm_csr(csr.indptr, csr.indices, graph.in_weights(w), h, graph.num_nodes(), mean=True,
                           tiles=graph.in_tiles())
        y, z, stats = sage_linear_fwd(h, ahn, weight, bias, gamma, beta, eps, relu, need_grad)
        ctx.graph, ctx.w, ctx.relu, ctx.use_pp = graph, w, relu, use_pp
        ctx.has_bias, ctx.has_ln = bias is not None, gamma is not None
        ctx.save_for_backward(h, ahn, weight, gamma, beta, z, stats, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, ahn, weight, gamma, beta, z, stats, y = ctx.saved_tensors
        need_h, need_w, need_b, need_g, need_be = ctx.needs_input_grad[:5]
        n_out, f = weight.shape[0], h.shape[1]
        dev = h.device
        dgamma = torch.empty(n_out, dtype=torch.float32, device=dev) if ctx.has_ln else None
        dbeta = torch.empty(n_out, dtype=torch.float32, device=dev) if ctx.has_ln else None
        dbias = torch.empty(n_out, dtype=torch.float32, device=dev) if ctx.has_bias else None
        # (1) LayerNorm/ReLU backward (+ bias/gamma/beta column sums); relu without LN masks on y
        dz = ln_relu_bwd(dy, z if ctx.has_ln else y, stats, gamma, beta, ctx.relu, dgamma, dbeta, dbias)
        # (2) dW = dZ^T [h | ahn]  (reduction over the nodes, split-K inside)
        dweight = None
        if need_w:
            dweight = torch.empty_like(weight)
            sage_linear_dw(dz, h, ahn, dweight)
        # (3) dh = dZ W_self + A_w^T (norm * (dZ W_neigh))
        dh = None
        if need_h:
            dh = gemm(dz, weight[:, :f])
            if ahn is not None:
                dahn = gemm(dz, weight[:, f:])
                g = ctx.graph
                rcsr = g.out_csr()
                spmm_csr(rcsr.indptr, rcsr.indices, g.out_weights(ctx.w, True), dahn, g.num_nodes(), out=dh,
                         accumulate=True, tiles=g.out_tiles())
        return dh, dweight, dbias, dgamma, dbeta, None, None, None, None, None


class _NarrowSageLayer(torch.autograd.Function):
    """The class-count-wide output layer in transform-then-aggregate form (csrc/narrow_layer.hip):
    logits = h W_s^T + b + mean-aggregate(h W_n^T) -- equal by linearity to [h | norm*A_w h] W^T + b."""

    @staticmethod
    def forward(ctx, h, weight, bias, graph, w):
        lib = _lib.load()
        h = _row_major(h)
        n, f = h.shape
        c = weight.shape[0]
        y = torch.empty((n, c), dtype=torch.float32, device=h.device)
        tn = torch.empty((n, c), dtype=torch.float32, device=h.device)
        check(lib.gte_sage_narrow_fwd(ptr(h), _ld(h), f, ptr(weight), _ld(weight), ptr(bias), c, ptr(y), c, ptr(tn), c, n,
                                      current_stream()), "gte_sage_narrow_fwd")
        csr = graph.in_csr()
        spmm_csr(csr.indptr, csr.indices, graph.in_weights(w), tn, n, mean=True, out=y, accumulate=True)
        ctx.graph, ctx.w = graph, w
        ctx.save_for_backward(h, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        h, weight = ctx.saved_tensors
        g = ctx.graph
        n, f = h.shape
        c = weight.shape[0]
        dy = _row_major(dy.contiguous())
        rcsr = g.out_csr()
        q = spmm_csr(rcsr.indptr, rcsr.indices, g.out_weights(ctx.w, True), dy, n)
        dh = torch.empty((n, f), dtype=torch.float32, device=h.device) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight)
        db = torch.empty(c, dtype=torch.float32, device=h.device)
        ws = _workspace(lib.gte_sage_narrow_bwd_workspace_bytes(n, f, c), h.device, "narrow")
        check(lib.gte_sage_narrow_bwd(ptr(dy), _ld(dy), ptr(q), _ld(q), ptr(h), _ld(h), f, ptr(weight), _ld(weight), c,
                                      ptr(dh), f, ptr(dw), _ld(dw), ptr(db), n, ptr(ws), ws.numel(), current_stream()),
              "gte_sage_narrow_bwd")
        return dh, dw, db, None, None


def sage_layer(graph, h, weight, bias=None, gamma=None, beta=None, edge_weight=None, relu: bool = False,
               eps: float = 1e-5, use_pp: bool = False) -> torch.Tensor:
    require_device(h, "sage_layer")
    if (gamma is None and not relu and not use_pp and bias is not None and h.dim() == 2
            and _lib.load().gte_sage_narrow_supported(h.shape[1], weight.shape[0])):
        return _NarrowSageLayer.apply(h, weight, bias, graph, edge_weight)
    if (not use_pp and gamma is not None and bias is not None and h.dim() == 2 and h.shape[1] > weight.shape[0]
            and not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (h, weight, bias, gamma, beta)))):
        y = _sage_layer_transform_first(graph, h, weight, bias, gamma, beta, edge_weight, relu, eps)
        if y is not None:
            return y
    return _SageLayer.apply(h, weight, bias, gamma, beta, graph, edge_weight, relu, eps, use_pp)


def _sage_layer_transform_first(graph, h, weight, bias, gamma, beta, edge_weight, relu, eps):
    """Inference-only forward of a narrowing layer (831 -> 256) in transform-then-aggregate order:
    z = h W_s^T + b + mean-aggregate(h W_n^T) -- the aggregation moves out_feats columns instead of in_feats --
    with LayerNorm(+ReLU) in the aggregation's epilogue.  Same math as models.py:53-72 by linearity (the step engine
    trains this way too); used when nothing requires grad (eval / predict).  None -> caller takes the general path."""
    lib = _lib.load()
    h = _row_major(h)
    n, fin = h.shape
    fout = weight.shape[0]
    if h.dtype != torch.float32:
        return None
    t = torch.empty((n, 2 * fout), dtype=torch.float32, device=h.device)
    check(lib.gte_sage_transform_fwd(ptr(h), _ld(h), fin, ptr(weight), _ld(weight), ptr(bias), fout, ptr(t), 2 * fout, n,
                                     current_stream()), "gte_sage_transform_fwd")
    csr = graph.in_csr()
    w = graph.in_weights(edge_weight)
    y = torch.empty((n, fout), dtype=torch.float32, device=h.device)
    # (unpadded rows here: the fused kernel's 16-byte chunks need fout % 4 == 0)
    if fout % 4 == 0 and lib.gte_spmm_csr_accumulate_ln_supported(fout) and not use_tiled(n, fout, csr.indices.numel(), fused_ln=True):
        check(lib.gte_spmm_csr_accumulate_ln(ptr(csr.indptr), ptr(csr.indices), ptr(w), ptr(t) + 4 * fout, 2 * fout, ptr(t),
                                             2 * fout, n, fout, _lib.REDUCE_MEAN, ptr(gamma), ptr(beta), float(eps), int(relu),
                                             ptr(y), fout, None, current_stream()), "gte_spmm_csr_accumulate_ln")
        return y
    # large graphs: LDS-staged aggregation (no LayerNorm epilogue there), then LayerNorm
    spmm_csr(csr.indptr, csr.indices, w, t[:, fout:], n, mean=True, out=t[:, :fout], accumulate=True,
             tiles=graph.in_tiles() if use_tiled(n, fout, csr.indices.numel(), fused_ln=True) else None)
    check(lib.gte_ln_relu_fwd(ptr(t), 2 * fout, ptr(gamma), ptr(beta), float(eps), int(relu), ptr(y), fout, None, n, fout,
                              current_stream()), "gte_ln_relu_fwd")
    return y


class _WeightedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, class_weight):
        out3, dl = weighted_ce(logits, labels, class_weight, want_grad=logits.requires_grad)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(out3)
        return out3[0], out3

    @staticmethod
    def backward(ctx, dloss, _d3):
        (dl,) = ctx.saved_tensors
        return dl * dloss, None, None


def cross_entropy(logits, labels, class_weight=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(loss, out3) -- ``nn.CrossEntropyLoss(weight)(logits, labels.long())`` (model_train.py:171,327)."""
    return _WeightedCE.apply(logits, labels, class_weight)


GEMM_F32, GEMM_SPLIT_BF16 = 0, 1


def set_gemm_mode(mode) -> int:
    """Arithmetic of the fp32 transform GEMMs, process-wide (gte_gemm_set_mode): ``"f32"`` (v_mfma_f32_32x32x2_f32, the
    default) or ``"split_bf16"`` (three exact bf16 pieces per operand, six bf16 MFMA products, fp32 accumulation: fp32 in and
    out at fp32 accuracy, csrc/gemm_split.h).  Returns the previous mode."""
    lib = _lib.load()
    code = {"f32": GEMM_F32, "split_bf16": GEMM_SPLIT_BF16, "split": GEMM_SPLIT_BF16}.get(mode, mode)
    prev = lib.gte_gemm_get_mode()
    check(lib.gte_gemm_set_mode(int(code)), "gte_gemm_set_mode")
    return prev


def get_gemm_mode() -> int:
    return _lib.load().gte_gemm_get_mode()


# --------------------------------------------------------------------------------------------------
# P3 operands (three bf16 planes of an fp32 matrix, csrc/p3.h) and the planes GEMMs
# --------------------------------------------------------------------------------------------------
class P3:
    """Device image of a logical fp32 matrix [rows][cols] in the P3 format.  Row-major (the default): ``data`` is a uint8 tensor
    [rows_cap, ldp], 16-feature block fb of row r at r * ldp + 96 fb.  BLOCK-MAJOR (``block_major``; weight images, the B operand
    of the NT planes GEMMs): ``data`` is [blocks, rows_cap, 96], block fb is one contiguous [rows_cap][96 bytes] run -- what the
    block-major-weights kernel streams straight into registers (csrc/gemm_p3.hip: gemm_p3_nt_sq_kernel).  The C ABI takes one
    signed stride per image: ``ldp`` > 0 = row stride of a row-major image, < 0 = minus the block stride of a block-major one."""
    __slots__ = ("data", "rows", "cols", "row_map", "res_rows", "block_major")

    def __init__(self, data: torch.Tensor, rows: int, cols: int, row_map: Optional[torch.Tensor] = None, res_rows: int = 0,
                 block_major: bool = False):
        """``row_map`` (int32, rows rounded up to 16, plus 1, entries; the entries past ``rows`` = res_rows): the matrix is the
        rows row_map[0 .. rows) of the RESIDENT image ``data`` [res_rows] -- what ResidentPages hands the input layer instead of
        a per-batch copy (gte_gemm_p3_nt_rows / gte_gemm_p3_tn_rows)."""
        self.data, self.rows, self.cols, self.row_map, self.res_rows, self.block_major = data, rows, cols, row_map, res_rows, block_major

    @property
    def ldp(self) -> int:
        return -self.data.stride(0) if self.block_major else self.data.stride(0)

    def at(self, row: int = 0, block: int = 0) -> int:
        """device address of 16-feature block ``block`` of row ``row`` (sub-images: the halves of a weight arrangement)"""
        if self.block_major:
            return self.data.data_ptr() + block * self.data.stride(0) + row * 96
        return self.data.data_ptr() + row * self.data.stride(0) + block * 96

    @staticmethod
    def empty(rows: int, cols: int, device, rows_cap: Optional[int] = None, block_major: bool = False) -> "P3":
        ldp = _lib.load().gte_p3_row_bytes(cols)
        if block_major:
            return P3(torch.empty((max(ldp // 96, 1), max(rows_cap or rows, 1), 96), dtype=torch.uint8, device=device), rows, cols,
                      block_major=True)
        return P3(torch.empty((max(rows_cap or rows, 1), ldp), dtype=torch.uint8, device=device), rows, cols)

    def view_rows(self, rows: int) -> "P3":
        return P3(self.data, rows, self.cols, self.row_map, self.res_rows, self.block_major)

    def gathered(self) -> "P3":
        """a contiguous copy of a row-mapped image (tests)"""
        if self.row_map is None:
            return self
        return P3(self.data[self.row_map[:self.rows].long()].contiguous(), self.rows, self.cols)


def p3_from_f32(src: torch.Tensor, transpose: bool = False, out: Optional[P3] = None, row0: int = 0, block_major: bool = False,
                block0: int = 0) -> P3:
    """P3 image of ``src`` ([rows][cols], or its transpose).  ``out``/``row0``/``block0``: write rows [row0, row0 + rows) from column
    block ``block0`` on of an existing image (how the weight arrangements [W_s ; W_n] and [W_s | W_n] are put together).
    ``block_major``: a new image in the block-major layout (weights; see :class:`P3`)."""
    require_device(src, "p3_from_f32")
    lib = _lib.load()
    src = _row_major(src)
    rows, cols = (src.shape[1], src.shape[0]) if transpose else src.shape
    if out is None:
        out = P3.empty(rows, cols, src.device, block_major=block_major)
    check(lib.gte_p3_from_f32(ptr(src), _ld(src), rows, cols, int(transpose), out.at(row0, block0), out.ldp,
                              current_stream()), "gte_p3_from_f32")
    return out


def p3_to_f32(img: P3) -> torch.Tensor:
    lib = _lib.load()
    img = img.gathered()
    out = torch.empty((img.rows, img.cols), dtype=torch.float32, device=img.data.device)
    check(lib.gte_p3_to_f32(ptr(img.data), img.ldp, img.rows, img.cols, ptr(out), max(img.cols, 1), current_stream()), "gte_p3_to_f32")
    return out


def gemm_p3_nt(a1: P3, b: P3, a2: Optional[P3] = None, bias: Optional[torch.Tensor] = None, bias_cols: int = 0,
               out: Optional[torch.Tensor] = None, relu: bool = False, accumulate: bool = False) -> torch.Tensor:
    """out[m, n] (+)= [a1 | a2] b^T (+ bias): gte_gemm_p3_nt"""
    lib = _lib.load()
    m, n = a1.rows, b.rows
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a1.data.device)
    if a1.row_map is not None and a2 is not None:
        if a2.row_map is None or a2.cols != a1.cols or a2.res_rows != a1.res_rows:
            raise ValueError("gemm_p3_nt: two K segments behind a row map need two resident images of the same shape")
        with _timed("gemm_nt", 2.0 * m * n * 2 * a1.cols):
            check(lib.gte_gemm_p3_nt_rows2(ptr(a1.data), a1.ldp, ptr(a2.data), a2.ldp, a1.cols, ptr(a1.row_map), a1.res_rows, ptr(b.data),
                                           b.ldp, ptr(bias), bias_cols, ptr(out), _ld(out), m, n, int(relu), int(accumulate),
                                           current_stream()), "gte_gemm_p3_nt_rows2")
        return out
    if a1.row_map is not None:
        with _timed("gemm_nt", 2.0 * m * n * a1.cols):
            check(lib.gte_gemm_p3_nt_rows(ptr(a1.data), a1.ldp, a1.cols, ptr(a1.row_map), a1.res_rows, ptr(b.data), b.ldp, ptr(bias),
                                          bias_cols, ptr(out), _ld(out), m, n, int(relu), int(accumulate), current_stream()),
                  "gte_gemm_p3_nt_rows")
        return out
    with _timed("gemm_nt", 2.0 * m * n * (a1.cols + (a2.cols if a2 is not None else 0))):
        check(lib.gte_gemm_p3_nt(ptr(a1.data), a1.ldp, a1.cols, ptr(a2.data) if a2 is not None else None,
                                 a2.ldp if a2 is not None else 0, a2.cols if a2 is not None else 0, ptr(b.data), b.ldp, ptr(bias),
                                 bias_cols, ptr(out), _ld(out), m, n, int(relu), int(accumulate), current_stream()), "gte_gemm_p3_nt")
    return out


def gemm_p3_nt_ln_fwd(a1: P3, b: P3, a2: Optional[P3], bias, gamma, beta, eps: float, relu: bool, z: torch.Tensor,
                      y: Optional[torch.Tensor] = None, yp3: Optional[P3] = None, stats: Optional[torch.Tensor] = None):
    """z = [a1 | a2] b^T + bias and its LayerNorm(+ReLU) in one launch (gte_gemm_p3_nt_ln_fwd / ..._rows2_ln_fwd for two row-mapped
    resident images): z fp32, stats [2 m], y fp32 and / or P3 image."""
    lib = _lib.load()
    m, n = a1.rows, b.rows
    common = (ptr(bias), ptr(gamma), ptr(beta), float(eps), int(relu), ptr(z), _ld(z), ptr(y), 0 if y is None else _ld(y),
              ptr(yp3.data) if yp3 is not None else None, yp3.ldp if yp3 is not None else 0, ptr(stats), m, n, current_stream())
    if a1.row_map is not None:
        if a2 is None or a2.row_map is None:
            raise ValueError("gemm_p3_nt_ln_fwd: row-mapped operands come as two resident images")
        check(lib.gte_gemm_p3_nt_rows2_ln_fwd(ptr(a1.data), a1.ldp, ptr(a2.data), a2.ldp, a1.cols, ptr(a1.row_map), a1.res_rows, ptr(b.data),
                                              b.ldp, *common), "gte_gemm_p3_nt_rows2_ln_fwd")
    else:
        check(lib.gte_gemm_p3_nt_ln_fwd(ptr(a1.data), a1.ldp, a1.cols, ptr(a2.data) if a2 is not None else None,
                                        a2.ldp if a2 is not None else 0, a2.cols if a2 is not None else 0, ptr(b.data), b.ldp, *common),
              "gte_gemm_p3_nt_ln_fwd")
    return z


def gemm_p3_nt_ln_bwd(a1: P3, b: P3, a2: Optional[P3], z: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                      relu: bool, dz: Optional[torch.Tensor], dzp3: Optional[P3] = None, dgamma=None, dbeta=None, dbias=None) -> torch.Tensor:
    """dz = LN'(z)(mask . ([a1 | a2] b^T)) without storing the product (gte_gemm_p3_nt_ln_bwd): the backward GEMM that produces
    d(loss)/d(y of the layer below) with that layer's LayerNorm(+ReLU) backward as its epilogue."""
    lib = _lib.load()
    m, n = a1.rows, b.rows
    ws = _workspace(lib.gte_gemm_p3_nt_ln_bwd_workspace_bytes(m, n), a1.data.device, "gemm_p3_lnb")
    check(lib.gte_gemm_p3_nt_ln_bwd(ptr(a1.data), a1.ldp, a1.cols, ptr(a2.data) if a2 is not None else None,
                                    a2.ldp if a2 is not None else 0, a2.cols if a2 is not None else 0, ptr(b.data), b.ldp, ptr(z),
                                    _ld(z), ptr(stats), ptr(gamma), ptr(beta), int(relu), ptr(dz), _ld(dz) if dz is not None else 0,
                                    ptr(dzp3.data) if dzp3 is not None else None, dzp3.ldp if dzp3 is not None else 0, ptr(dgamma),
                                    ptr(dbeta), ptr(dbias), m, n, ptr(ws), ws.numel(), current_stream()), "gte_gemm_p3_nt_ln_bwd")
    return dz


def gemm_p3_nt_smallk_bwd(a1: P3, b: P3, a2: Optional[P3], x, ahn, weight, bias, gamma, beta, stats, relu: bool, dW, dbias=None,
                          dgamma=None, dbeta=None):
    """The backward GEMM that produces d(loss)/d(y) of a short-input INPUT layer with that layer's whole backward as its epilogue
    (gte_gemm_p3_nt_smallk_bwd): the product is not stored, dW = dz^T [x | ahn] and the LayerNorm parameter gradients come out."""
    lib = _lib.load()
    m, n = a1.rows, b.rows
    x, weight = _row_major(x), _row_major(weight)
    k1 = x.shape[1]
    k2 = 0
    if ahn is not None:
        ahn = _row_major(ahn)
        k2 = ahn.shape[1]
    ws = _workspace(lib.gte_gemm_p3_nt_smallk_bwd_workspace_bytes(m, k1 + k2, n), a1.data.device, "gemm_p3_skb")
    check(lib.gte_gemm_p3_nt_smallk_bwd(ptr(a1.data), a1.ldp, a1.cols, ptr(a2.data) if a2 is not None else None,
                                        a2.ldp if a2 is not None else 0, a2.cols if a2 is not None else 0, ptr(b.data), b.ldp, ptr(x),
                                        _ld(x), k1, ptr(ahn), 0 if ahn is None else _ld(ahn), k2, ptr(weight), _ld(weight), ptr(bias),
                                        ptr(gamma), ptr(beta), ptr(stats), int(relu), ptr(dW), _ld(dW), ptr(dbias), ptr(dgamma),
                                        ptr(dbeta), m, n, ptr(ws), ws.numel(), current_stream()), "gte_gemm_p3_nt_smallk_bwd")
    return dW


def gemm_p3_tn(a: P3, b: P3, a2: Optional[P3] = None, b2: Optional[P3] = None, two_segments: bool = False,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m, n] = a^T b over the rows (two_segments: out = [a^T b | a2^T b2], a2 / b2 default to a / b): gte_gemm_p3_tn"""
    lib = _lib.load()
    m, k = a.cols, a.rows
    nseg = b.cols if two_segments else 0
    n = 2 * b.cols if two_segments else b.cols
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.data.device)
    ws = _workspace(lib.gte_gemm_p3_tn_workspace_bytes(m, n, nseg, k), a.data.device, "gemm_p3")
    if b.row_map is not None and b2 is not None:
        if a2 is not None or not two_segments or b2.row_map is None or b2.res_rows != b.res_rows:
            raise ValueError("gemm_p3_tn: two resident images behind a row map: out = [a^T b | a^T b2]")
        with _timed("gemm_tn", 2.0 * m * n * k):
            check(lib.gte_gemm_p3_tn_rows2(ptr(a.data), a.ldp, ptr(b.data), b.ldp, ptr(b2.data), b2.ldp, ptr(b.row_map), b.res_rows, nseg,
                                           ptr(out), _ld(out), m, n, k, ptr(ws), ws.numel(), current_stream()), "gte_gemm_p3_tn_rows2")
        return out
    if b.row_map is not None:
        with _timed("gemm_tn", 2.0 * m * n * k):
            check(lib.gte_gemm_p3_tn_rows(ptr(a.data), a.ldp, ptr(a2.data) if a2 is not None else None, a2.ldp if a2 is not None else 0,
                                          ptr(b.data), b.ldp, ptr(b.row_map), b.res_rows, nseg, ptr(out), _ld(out), m, n, k, ptr(ws),
                                          ws.numel(), current_stream()), "gte_gemm_p3_tn_rows")
        return out
    with _timed("gemm_tn", 2.0 * m * n * k):
        check(lib.gte_gemm_p3_tn(ptr(a.data), a.ldp, ptr(a2.data) if a2 is not None else None, a2.ldp if a2 is not None else 0,
                                 ptr(b.data), b.ldp, ptr(b2.data) if b2 is not None else None, b2.ldp if b2 is not None else 0,
                                 nseg, ptr(out), _ld(out), m, n, k, ptr(ws), ws.numel(), current_stream()), "gte_gemm_p3_tn")
    return out
