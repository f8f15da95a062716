"""Multi-head graph attention node classifier on the HIP path (BASELINE.json configs[2]: "4-head GAT bf16").

The reference repository contains no GAT; this follows the standard formulation with DGL
``GATConv`` semantics (LeakyReLU 0.2, softmax over in-edges, hidden layers concatenate heads, the output
layer averages them).  Its oracle is ``oracle/gat_cpu.py`` -- parity unpinned.  ``gather_dtype=torch.bfloat16``
stores the projected features that the aggregation gathers in bf16 (fp32 accumulate, fp32 everything else).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib, ops


class _Linear(torch.autograd.Function):
    """y = x W^T through the fp32 MFMA GEMM (and its dX / dW)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.gemm(x, w, trans_b=True)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gemm(dy, w) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(dy, x, trans_a=True) if ctx.needs_input_grad[1] else None
        return dx, dw


class _GatAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, a_l, a_r, bias, graph, heads, bf16):
        lib, P, st = _lib.load(), _lib.ptr, _lib.current_stream()
        z = ops._row_major(z)
        n, hd = z.shape
        dim = hd // heads
        dev = z.device
        csr = graph.in_csr()
        el = torch.empty((n, heads), dtype=torch.float32, device=dev)
        er = torch.empty_like(el)
        zb = torch.empty((n, hd), dtype=torch.bfloat16, device=dev) if bf16 else None
        _lib.check(lib.gte_gat_scores(P(z), ops._ld(z), P(a_l), P(a_r), P(el), P(er), P(zb), hd, n, heads, dim, st),
                   "gte_gat_scores")
        zg = zb if bf16 else z
        out = torch.empty((n, hd), dtype=torch.float32, device=dev)
        smax, ssum = torch.empty_like(el), torch.empty_like(el)
        _lib.check(lib.gte_gat_aggregate_fwd(P(csr.indptr), P(csr.indices), P(zg), ops._ld(zg),
                                             _lib.GTE_BF16 if bf16 else _lib.GTE_F32, P(el), P(er), P(bias), P(out), hd,
                                             P(smax), P(ssum), n, heads, dim, st), "gte_gat_aggregate_fwd")
        ctx.graph, ctx.heads, ctx.bf16, ctx.has_bias = graph, heads, bf16, bias is not None
        ctx.save_for_backward(z, zg, a_l, a_r, el, er, smax, ssum)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib, P, st = _lib.load(), _lib.ptr, _lib.current_stream()
        z, zg, a_l, a_r, el, er, smax, ssum = ctx.saved_tensors
        g, heads = ctx.graph, ctx.heads
        n, hd = z.shape
        dim = hd // heads
        dev = z.device
        dout = ops._row_major(dout.contiguous())
        csr, rcsr = g.in_csr(), g.out_csr()
        e = csr.indices.numel()
        ds = torch.empty((max(e, 1), heads), dtype=torch.float32, device=dev)
        der, dele = torch.empty((n, heads), dtype=torch.float32, device=dev), torch.empty((n, heads), dtype=torch.float32, device=dev)
        dz = torch.empty((n, hd), dtype=torch.float32, device=dev)
        da_l, da_r = torch.empty_like(a_l), torch.empty_like(a_r)
        dbias = torch.empty(hd, dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = ops._workspace(lib.gte_gat_bwd_workspace_bytes(n, heads, dim), dev, "gat")
        _lib.check(lib.gte_gat_aggregate_bwd(P(csr.indptr), P(csr.indices), P(rcsr.indptr), P(rcsr.indices),
                                             P(g.out_to_in_pos()), P(zg), ops._ld(zg),
                                             _lib.GTE_BF16 if ctx.bf16 else _lib.GTE_F32, P(z), ops._ld(z), P(el), P(er),
                                             P(smax), P(ssum), P(a_l), P(a_r), P(dout), ops._ld(dout), P(ds), P(der),
                                             P(dele), P(dz), hd, P(da_l), P(da_r), P(dbias), n, heads, dim, P(ws),
                                             ws.numel(), st), "gte_gat_aggregate_bwd")
        return dz, da_l, da_r, dbias, None, None, None


class GATLayer(nn.Module):
    def __init__(self, in_feats, out_feats, heads, mean_heads=False, bias=True, gather_dtype=torch.float32):
        super().__init__()
        self.heads, self.out_feats, self.mean_heads = heads, out_feats, mean_heads
        self.bf16 = gather_dtype == torch.bfloat16
        self.fc = nn.Parameter(torch.empty(heads * out_feats, in_feats))
        self.attn_l = nn.Parameter(torch.empty(heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(heads, out_feats))
        self.bias = nn.Parameter(torch.zeros(out_feats if mean_heads else heads * out_feats)) if bias else None
        gain = math.sqrt(2.0)
        nn.init.xavier_normal_(self.fc, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)

    def forward(self, g, x):
        _lib.require_device(x, "GATLayer")
        z = _Linear.apply(x, self.fc)
        agg_bias = None if self.mean_heads else self.bias
        out = _GatAggregate.apply(z, self.attn_l.reshape(-1), self.attn_r.reshape(-1), agg_bias, g, self.heads, self.bf16)
        if self.mean_heads:
            out = out.view(out.shape[0], self.heads, self.out_feats).mean(1)      # head average of the output layer
            if self.bias is not None:
                out = out + self.bias
        return out


class GAT(nn.Module):
    """n_layers GAT layers: hidden = heads x n_hidden concatenated + activation; output = mean over heads."""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers=3, heads=4, activation=F.elu, gather_dtype=torch.float32):
        super().__init__()
        self.activation = activation
        self.layers = nn.ModuleList()
        d = in_feats
        for _ in range(n_layers - 1):
            self.layers.append(GATLayer(d, n_hidden, heads, gather_dtype=gather_dtype))
            d = heads * n_hidden
        self.layers.append(GATLayer(d, n_classes, heads, mean_heads=True, gather_dtype=gather_dtype))

    def forward(self, g, x=None):
        h = g.ndata['feat'] if x is None else x
        for i, layer in enumerate(self.layers):
            h = layer(g, h)
            if i != len(self.layers) - 1:
                h = self.activation(h)
        return h
