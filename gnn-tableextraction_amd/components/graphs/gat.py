"""Multi-head graph attention node classifier on the HIP path (BASELINE.json configs[2]: "4-head GAT bf16").

The reference repository contains no GAT; this follows the standard formulation with DGL
``GATConv`` semantics (LeakyReLU 0.2, softmax over in-edges, hidden layers concatenate heads, the output
layer averages them).  Its oracle is ``oracle/gat_cpu.py`` -- PARITY UNPINNED.

Precision modes (``GAT(..., gather_dtype=, compute_dtype=)``):
  gather_dtype  = bfloat16   the projected features the aggregation gathers are stored in bf16 (fp32 accumulate)
  compute_dtype = bfloat16   the projection z = X W^T runs on ``v_mfma_f32_32x32x16_bf16`` with bf16 operands and fp32
                             accumulation (csrc/gemm_bf16.hip); X's bf16 copy is written by the PREVIOUS layer's aggregation
                             epilogue together with the ELU (no separate activation or cast pass); the output layer's mean
                             over heads + bias is that kernel's epilogue too.  The backward GEMMs (dX, dW) take fp32 operands and run in the split mode of
                             the fp32 GEMM (csrc/gemm_split.h: six bf16 MFMA products of exact bf16 pieces, fp32-accurate).
With both fp32 (the default) every step is fp32.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib, ops


def _bf16_copy(x: torch.Tensor, activation: int = 0) -> torch.Tensor:
    """bf16 [n, round_up(cols, 8)] copy of a fp32 matrix (zero padded), optionally through ELU: gte_cast_bf16."""
    x = ops._row_major(x)
    n, c = x.shape
    cp = -(-c // 8) * 8
    y = torch.empty((n, cp), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().gte_cast_bf16(_lib.ptr(x), ops._ld(x), _lib.ptr(y), cp, n, c, activation, _lib.current_stream()),
               "gte_cast_bf16")
    return y


class _Linear(torch.autograd.Function):
    """y = x W^T.  fp32 MFMA GEMM, or -- with a bf16 copy of x -- bf16 MFMA with fp32 accumulation; dX / dW in fp32 MFMA."""

    @staticmethod
    def forward(ctx, x, w, x_bf16, bf16):
        ctx.save_for_backward(x, w)
        ctx.bf16 = bool(bf16)
        if not bf16:
            return ops.gemm(x, w, trans_b=True)
        lib, P = _lib.load(), _lib.ptr
        xb = x_bf16 if x_bf16 is not None else _bf16_copy(x)
        wb = _bf16_copy(w)                                   # [out, round_up(in, 8)]
        n, out_f = x.shape[0], w.shape[0]
        z = torch.empty((n, out_f), dtype=torch.float32, device=x.device)
        _lib.check(lib.gte_gemm_bf16_nt(P(xb), xb.stride(0), P(wb), wb.stride(0), P(z), out_f, n, out_f, wb.shape[1],
                                        _lib.current_stream()), "gte_gemm_bf16_nt")
        return z

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        if not ctx.bf16:
            dx = ops.gemm(dy, w) if ctx.needs_input_grad[0] else None
            dw = ops.gemm(dy, x, trans_a=True) if ctx.needs_input_grad[1] else None
            return dx, dw, None, None
        # bf16 configuration: dX / dW on the bf16 matrix pipe too, through the split mode of the fp32 GEMM (three exact bf16
        # pieces per fp32 operand, fp32 accumulate: fp32-accurate results at 1.3-1.5x the fp32 MFMA rate) -- for this
        # thread's two launches only
        lib = _lib.load()
        _lib.check(lib.gte_gemm_set_thread_mode(ops.GEMM_SPLIT_BF16), "gte_gemm_set_thread_mode")
        try:
            dx = ops.gemm(dy, w) if ctx.needs_input_grad[0] else None
            dw = ops.gemm(dy, x, trans_a=True) if ctx.needs_input_grad[1] else None
        finally:
            lib.gte_gemm_set_thread_mode(-1)
        return dx, dw, None, None


class _GatAggregate(torch.autograd.Function):
    """Attention aggregation of one layer with its epilogue: out = act(aggregate + bias) [+ bf16 copy] for hidden layers,
    mean over heads + bias for the output layer."""

    @staticmethod
    def forward(ctx, z, a_l, a_r, bias, mean_bias, graph, heads, bf16_gather, elu, mean_heads, emit_bf16):
        lib, P, st = _lib.load(), _lib.ptr, _lib.current_stream()
        z = ops._row_major(z)
        n, hd = z.shape
        dim = hd // heads
        dev = z.device
        csr = graph.in_csr()
        el = torch.empty((n, heads), dtype=torch.float32, device=dev)
        er = torch.empty_like(el)
        zb = torch.empty((n, hd), dtype=torch.bfloat16, device=dev) if bf16_gather else None
        _lib.check(lib.gte_gat_scores(P(z), ops._ld(z), P(a_l), P(a_r), P(el), P(er), P(zb), hd, n, heads, dim, st),
                   "gte_gat_scores")
        zg = zb if bf16_gather else z
        smax, ssum = torch.empty_like(el), torch.empty_like(el)
        out = None if mean_heads else torch.empty((n, hd), dtype=torch.float32, device=dev)
        out_mean = torch.empty((n, dim), dtype=torch.float32, device=dev) if mean_heads else None
        out_b = torch.empty((n, hd), dtype=torch.bfloat16, device=dev) if (emit_bf16 and not mean_heads and hd % 8 == 0) else None
        _lib.check(lib.gte_gat_aggregate_fwd_ex(P(csr.indptr), P(csr.indices), P(zg), ops._ld(zg),
                                                _lib.GTE_BF16 if bf16_gather else _lib.GTE_F32, P(el), P(er), P(bias), P(out), hd,
                                                P(smax), P(ssum), n, heads, dim, int(elu), P(out_b), hd, P(out_mean), dim,
                                                P(mean_bias), st), "gte_gat_aggregate_fwd_ex")
        ctx.graph, ctx.heads, ctx.bf16, ctx.has_bias = graph, heads, bf16_gather, bias is not None
        ctx.elu, ctx.mean_heads, ctx.has_mean_bias = bool(elu), bool(mean_heads), mean_bias is not None
        res = out_mean if mean_heads else out
        ctx.save_for_backward(z, zg, a_l, a_r, el, er, smax, ssum, res if elu else None)
        if out_b is not None:
            ctx.mark_non_differentiable(out_b)
        return res, out_b

    @staticmethod
    def backward(ctx, dout, _dbf16):
        lib, P, st = _lib.load(), _lib.ptr, _lib.current_stream()
        z, zg, a_l, a_r, el, er, smax, ssum, act_out = ctx.saved_tensors
        g, heads = ctx.graph, ctx.heads
        n, hd = z.shape
        dim = hd // heads
        dev = z.device
        dout = ops._row_major(dout.contiguous())
        dmean_bias = dout.sum(0) if ctx.has_mean_bias else None
        # gradient w.r.t. the pre-epilogue aggregate (broadcast of the head mean / heads, ELU' from the saved output): formed
        # inside the backward's first kernel and left in `dfull` for the others (gte_gat_aggregate_bwd_ex)
        prep = ctx.elu or ctx.mean_heads
        dfull = torch.empty((n, hd), dtype=torch.float32, device=dev) if prep else None
        csr, rcsr = g.in_csr(), g.out_csr()
        e = csr.indices.numel()
        ds = torch.empty((max(e, 1), heads), dtype=torch.float32, device=dev)
        der, dele = torch.empty((n, heads), dtype=torch.float32, device=dev), torch.empty((n, heads), dtype=torch.float32, device=dev)
        dz = torch.empty((n, hd), dtype=torch.float32, device=dev)
        da_l, da_r = torch.empty_like(a_l), torch.empty_like(a_r)
        dbias = torch.empty(hd, dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = ops._workspace(lib.gte_gat_bwd_workspace_bytes(n, heads, dim), dev, "gat")
        _lib.check(lib.gte_gat_aggregate_bwd_ex(P(csr.indptr), P(csr.indices), P(rcsr.indptr), P(rcsr.indices),
                                                P(g.out_to_in_pos()), P(zg), ops._ld(zg),
                                                _lib.GTE_BF16 if ctx.bf16 else _lib.GTE_F32, P(z), ops._ld(z), P(el), P(er),
                                                P(smax), P(ssum), P(a_l), P(a_r), P(dout), ops._ld(dout),
                                                P(act_out) if ctx.elu else None, hd, int(ctx.mean_heads), P(dfull), hd,
                                                P(ds), P(der), P(dele), P(dz), hd, P(da_l), P(da_r), P(dbias), n, heads, dim,
                                                P(ws), ws.numel(), st), "gte_gat_aggregate_bwd_ex")
        return dz, da_l, da_r, dbias, dmean_bias, None, None, None, None, None, None


class GATLayer(nn.Module):
    def __init__(self, in_feats, out_feats, heads, mean_heads=False, bias=True, gather_dtype=torch.float32,
                 compute_dtype=torch.float32):
        super().__init__()
        self.heads, self.out_feats, self.mean_heads = heads, out_feats, mean_heads
        self.bf16 = gather_dtype == torch.bfloat16
        self.mfma_bf16 = compute_dtype == torch.bfloat16
        self.fc = nn.Parameter(torch.empty(heads * out_feats, in_feats))
        self.attn_l = nn.Parameter(torch.empty(heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(heads, out_feats))
        self.bias = nn.Parameter(torch.zeros(out_feats if mean_heads else heads * out_feats)) if bias else None
        gain = math.sqrt(2.0)
        nn.init.xavier_normal_(self.fc, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)

    def forward(self, g, x, x_bf16=None, fuse_elu=False, emit_bf16=False):
        """Returns (out, bf16 copy of out or None).  ``fuse_elu``: ELU in the aggregation's epilogue; ``x_bf16``: the bf16 copy
        of x a previous layer's epilogue wrote (bf16 projection only)."""
        _lib.require_device(x, "GATLayer")
        z = _Linear.apply(x, self.fc, x_bf16, self.mfma_bf16)
        agg_bias, mean_bias = (None, self.bias) if self.mean_heads else (self.bias, None)
        return _GatAggregate.apply(z, self.attn_l.reshape(-1), self.attn_r.reshape(-1), agg_bias, mean_bias, g, self.heads,
                                   self.bf16, fuse_elu, self.mean_heads, emit_bf16)


class GAT(nn.Module):
    """n_layers GAT layers: hidden = heads x n_hidden concatenated + activation; output = mean over heads."""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers=3, heads=4, activation=F.elu, gather_dtype=torch.float32,
                 compute_dtype=torch.float32):
        super().__init__()
        self.activation = activation
        self.mfma_bf16 = compute_dtype == torch.bfloat16
        self.layers = nn.ModuleList()
        d = in_feats
        for _ in range(n_layers - 1):
            self.layers.append(GATLayer(d, n_hidden, heads, gather_dtype=gather_dtype, compute_dtype=compute_dtype))
            d = heads * n_hidden
        self.layers.append(GATLayer(d, n_classes, heads, mean_heads=True, gather_dtype=gather_dtype, compute_dtype=compute_dtype))

    def forward(self, g, x=None):
        h = g.ndata['feat'] if x is None else x
        hb = None
        fuse = self.activation is F.elu                      # ELU runs in the aggregation's epilogue; anything else through torch
        for i, layer in enumerate(self.layers):
            last = i == len(self.layers) - 1
            h, hb = layer(g, h, hb, fuse_elu=fuse and not last, emit_bf16=self.mfma_bf16 and not last)
            if not last and not fuse:
                h, hb = self.activation(h), None
        return h
