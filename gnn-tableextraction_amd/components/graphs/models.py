"""GraphSAGE-GCN node classifier on the MI355X HIP path.

Drop-in for the reference module ``src/components/graphs/models.py`` (GcnSAGELayer :15-78,
GcnSAGE :80-116, WeightedMeanSAGELayer :118-152, MeanSAGE :154-170): same constructor
signatures, same attribute names (``layers[i].linear``, ``layers[i].lynorm``, ``dropout``), same
``state_dict`` keys, same parameter initialisation (and RNG consumption order, so a seed gives
the reference's weights), same ``model(g) -> logits[N, n_classes]`` call.  What changes is the
execution: per layer ONE autograd node whose forward and backward are HIP kernels behind the
C ABI -- CSR gather aggregation with the 1/in-degree norm folded in, a split-weight fp32-MFMA
GEMM instead of ``cat`` + ``Linear``, LayerNorm/ReLU kernels.  There is no CPU fallback.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops

_RELUS = (F.relu, torch.relu, torch.nn.functional.relu)


def _is_relu(act) -> bool:
    return act in _RELUS or isinstance(act, nn.ReLU)


class _Identity(nn.Module):
    """Stands where the reference keeps ``lambda x: x`` (use_lynorm=False): no parameters."""

    def forward(self, x):
        return x


class GcnSAGELayer(nn.Module):
    def __init__(self, in_feats, out_feats, activation, dropout, bias=True, use_pp=False, use_lynorm=True):
        super().__init__()
        # weight is [out, 2*in]: columns [0, in) multiply the node's own features, [in, 2*in) the
        # normalised neighbour sum (the cat order of models.py:69-72)
        self.linear = nn.Linear(2 * in_feats, out_feats, bias=bias)
        self.activation = activation
        self.use_pp = use_pp
        self.dropout = nn.Dropout(p=dropout) if dropout else 0.
        if use_lynorm:
            self.lynorm = nn.LayerNorm(out_feats, elementwise_affine=True)
        else:
            object.__setattr__(self, "lynorm", lambda x: x)
        self.in_feats, self.out_feats = in_feats, out_feats
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1. / math.sqrt(self.linear.weight.size(1))
        self.linear.weight.data.uniform_(-bound, bound)
        if self.linear.bias is not None:
            self.linear.bias.data.uniform_(-bound, bound)

    def forward(self, g, h, edge_weight=None):
        """``g``: PageGraph (or anything exposing in_csr()/out_csr()); ``h``: [N, in] (use_pp: [N, 2*in]).
        Edge weights come from ``g.edata['feat']`` as in the reference; a missing key means 1.0
        (the reference's ``--edge_features=False`` leaves it unset: loader.py:332)."""
        if edge_weight is None:
            edge_weight = g.edata.get("feat") if hasattr(g, "edata") else None
        ln = isinstance(self.lynorm, nn.LayerNorm)
        fused_relu = _is_relu(self.activation)
        if self.dropout and self.training:
            # dropout acts on cat(h, ah*norm) (models.py:60-61 of the reference: concat, then ONE dropout call over [N, 2 in]):
            # aggregate first, one mask over the concatenation, then the linear
            if not self.use_pp:
                ahn = ops.aggregate(g, h, edge_weight, mean=True)
                h = self.dropout(torch.cat((h, ahn), dim=1))
            else:
                h = self.dropout(h)
            out = ops.sage_layer(g, h, self.linear.weight, self.linear.bias,
                                 self.lynorm.weight if ln else None, self.lynorm.bias if ln else None,
                                 None, relu=fused_relu, eps=self.lynorm.eps if ln else 1e-5, use_pp=True)
        else:
            out = ops.sage_layer(g, h, self.linear.weight, self.linear.bias,
                                 self.lynorm.weight if ln else None, self.lynorm.bias if ln else None,
                                 edge_weight, relu=fused_relu, eps=self.lynorm.eps if ln else 1e-5,
                                 use_pp=self.use_pp)
        if self.activation and not fused_relu:
            out = self.activation(out)
        return out


class GcnSAGE(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout, use_pp=False):
        super().__init__()
        self.layers = nn.ModuleList()
        self.dropout = nn.Dropout(dropout)
        self.layers.append(GcnSAGELayer(in_feats, n_hidden, activation=activation, dropout=dropout,
                                        use_pp=use_pp, use_lynorm=True))
        for _ in range(n_layers - 2):
            self.layers.append(GcnSAGELayer(n_hidden, n_hidden, activation=activation, dropout=dropout,
                                            use_pp=False, use_lynorm=True))
        self.layers.append(GcnSAGELayer(n_hidden, n_classes, activation=None, dropout=False,
                                        use_pp=False, use_lynorm=False))

    def forward(self, g, node_feats=None, edge_index=None, edge_weight=None):
        """``model(g)`` as in the reference, or the tensor-level form
        ``model(None, node_feats, edge_index[, edge_weight])`` / ``model(node_feats, edge_index)``."""
        if torch.is_tensor(g):                       # model(node_feats, edge_index[, edge_weight])
            g, node_feats, edge_index, edge_weight = None, g, node_feats, edge_index
        if g is None:
            from ...graph import from_edge_index
            g = from_edge_index(edge_index, node_feats.shape[0], edge_weight)
        h = node_feats if node_feats is not None else g.ndata['feat']
        h = self.dropout(h)
        for layer in self.layers:
            h = layer(g, h)
        return h


class WeightedMeanSAGELayer(nn.Module):
    """Linear(cat(h, mean over in-edges of w_e * h[u]))  (reference models.py:118-152)."""

    def __init__(self, in_feat, out_feat):
        super().__init__()
        self.linear = nn.Linear(in_feat * 2, out_feat)

    def forward(self, g, h, w):
        return ops.sage_layer(g, h, self.linear.weight, self.linear.bias, None, None, w, relu=False)


class MeanSAGE(nn.Module):
    def __init__(self, in_feats, h_feats, num_classes, n_layers):
        super().__init__()
        self.n_layers = n_layers
        self.layers = nn.ModuleList([WeightedMeanSAGELayer(in_feats, h_feats)])
        for _ in range(n_layers - 1):
            self.layers.append(WeightedMeanSAGELayer(h_feats, h_feats))
        self.layers.append(WeightedMeanSAGELayer(h_feats, num_classes))

    def forward(self, g, h, w):
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            h = layer(g, h, w)
            if i != last:
                h = F.normalize(F.relu(h))
        return h
