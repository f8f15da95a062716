"""CPU oracle for the GAT layer of BASELINE cfg3  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference repository contains NO GAT (no attention, no bf16; ``SAGEConv`` is imported
but unused at src/components/graphs/models.py:1).  BASELINE.json configs[2] nevertheless names a "4-head GAT
bf16", so the layer is defined here by the standard formulation (Velickovic et al. 2018, as implemented by
DGL's GATConv):
    z = X W                         [N, H, D]
    e_uv,h = LeakyReLU_0.2( <a_l[h], z[u,h]> + <a_r[h], z[v,h]> )      for every edge u -> v
    alpha_uv,h = softmax over the in-edges of v
    out[v,h] = sum_u alpha_uv,h z[u,h]  + bias
    hidden layers concatenate the heads, the output layer averages them.
PARITY UNPINNED: there is no reference code, golden vector or published number for this row; this plain
PyTorch-CPU restatement (per-edge tensors + index_add, fp32 or fp64) is the only oracle.
"""
from __future__ import annotations

import torch


def bf16_round(t):
    """round to nearest even to bf16, kept in the working dtype"""
    return t.to(torch.bfloat16).to(t.dtype)


class _ProjectionBf16(torch.autograd.Function):
    """z = bf16(x) bf16(w)^T with wide accumulation -- the device's bf16 MFMA projection; its backward multiplies the
    UNROUNDED operands (components/graphs/gat.py::_Linear.backward keeps x and w in fp32)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return bf16_round(x) @ bf16_round(w).t()

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        return dz @ w, dz.t() @ x


def gat_layer(x, w, a_l, a_r, bias, src, dst, n, heads, mean_heads=False, slope=0.2, round_gather=False, round_proj=False):
    """x [N,F]; w [H*D, F]; a_l, a_r [H, D]; bias [H*D] (or [D] when mean_heads); src/dst int64 [E].
    round_proj / round_gather restate the device's bf16 configuration at ITS rounding points: the projection's operands
    (compute_dtype = bf16) and the projected features the aggregation gathers (gather_dtype = bf16; the attention scores are
    formed from the unrounded z, the backward sees the rounded values where the forward used them).  Against this form only
    the summation order differs."""
    z = (_ProjectionBf16.apply(x, w) if round_proj else x @ w.t()).view(n, heads, -1)   # [N,H,D]
    el = (z * a_l.unsqueeze(0)).sum(-1)                                   # [N,H]
    er = (z * a_r.unsqueeze(0)).sum(-1)
    e = torch.nn.functional.leaky_relu(el[src] + er[dst], slope)          # [E,H]
    m = torch.full((n, heads), -float("inf"), dtype=x.dtype).scatter_reduce(0, dst[:, None].expand(-1, heads), e,
                                                                            reduce="amax", include_self=True)
    p = torch.exp(e - m[dst])
    denom = torch.zeros(n, heads, dtype=x.dtype).index_add_(0, dst, p)
    alpha = p / denom[dst]
    zg = z + (bf16_round(z) - z).detach() if round_gather else z          # the gathered copy: rounded values, identity gradient
    out = torch.zeros(n, heads, z.shape[-1], dtype=x.dtype).index_add_(0, dst, alpha.unsqueeze(-1) * zg[src])
    if mean_heads:
        out = out.mean(1)
        return out + bias if bias is not None else out
    out = out.reshape(n, -1)
    return out + bias if bias is not None else out


def gat_forward(params, src, dst, n, x, heads, activation=torch.nn.functional.elu, round_gather=False, round_proj=False):
    """params: list of dicts(w, a_l, a_r, bias) ; hidden layers concat + activation, last layer mean of heads."""
    h = x
    for i, p in enumerate(params):
        last = i == len(params) - 1
        h = gat_layer(h, p["w"], p["a_l"], p["a_r"], p["bias"], src, dst, n, heads, mean_heads=last,
                      round_gather=round_gather, round_proj=round_proj)
        if not last:
            h = activation(h)
    return h
