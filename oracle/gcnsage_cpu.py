"""CPU oracle for the GcnSAGE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The shipped path (package
``gnn-tableextraction_amd``) never does; it fails loudly without the HIP library.

What it restates (reference file:line, relative to the upstream repository):

* ``src/components/graphs/models.py:15-78``   GcnSAGELayer  (norm, weighted-sum
  aggregation, concat, Linear, LayerNorm, activation)
* ``src/components/graphs/models.py:80-116``  GcnSAGE       (layer stack)
* ``src/components/graphs/models.py:118-170`` WeightedMeanSAGELayer / MeanSAGE
* ``src/models/model_train.py:168-171,320-332`` one optimisation step
  (CrossEntropyLoss(weight) -> backward -> Adam with L2-coupled weight decay)
* ``src/components/features/utils.py:71-101`` get_in_feats_ / calculate_hidden

The arithmetic of the aggregation itself lives in DGL (third-party, un-vendored,
un-pinned; call site ``models.py:53-54``: ``update_all(u_mul_e('h','feat','m'),
sum('m','h'))``).  It is restated here from DGL's published gSpMM semantics:
``out[v] = sum over in-edges e=(u->v) of w_e * h[u]``, 0 for in-degree-0 nodes,
duplicate edges accumulate.

Parity pinning: the reference holds no golden vectors for this path.  The oracle
is pinned against outputs of the reference's own ``models.py`` executed in the
build container under a stub ``dgl`` (``oracle/make_golden.py`` -> fixtures in
``tests/golden``), and against an independent fp64 dense-adjacency formulation
(``dense_reference_forward`` below).  The DGL kernel itself could not be run
(DGL is not installable here): for that one operator parity is "unpinned" beyond
its documented semantics.

Everything is plain PyTorch-CPU / numpy.  The aggregation goes through a CSR
SpMM (optionally the OpenMP C kernel in ``oracle/spmm_csr_omp.c`` -- the
row-parallel algorithm DGL's CPU backend uses), never through scatter/index_add.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------- #
# graph helpers (COO -> in-edge CSR, out-edge CSR)
# --------------------------------------------------------------------------- #


def coo_to_in_csr(src: np.ndarray, dst: np.ndarray, num_nodes: int,
                  weight: Optional[np.ndarray] = None):
    """Stable sort of the COO by destination -> (indptr, indices=src, w, perm).

    Row v of the CSR lists the sources of v's incoming edges in original edge
    order (stable), which fixes the summation order.
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    perm = np.argsort(dst, kind="stable")
    counts = np.bincount(dst, minlength=num_nodes)
    indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    w = None if weight is None else np.asarray(weight, dtype=np.float32)[perm]
    return indptr.astype(np.int32), src[perm].astype(np.int32), w, perm


def in_degree_norm(indptr: np.ndarray) -> np.ndarray:
    """models.py:74-78: 1/in_degree with inf -> 0, float32, shape [N,1]."""
    deg = np.diff(indptr.astype(np.int64)).astype(np.float32)
    with np.errstate(divide="ignore"):
        norm = np.float32(1.0) / deg
    norm[np.isinf(norm)] = 0.0
    return norm.reshape(-1, 1).astype(np.float32)


# --------------------------------------------------------------------------- #
# optional OpenMP CSR kernel (oracle/spmm_csr_omp.c)
# --------------------------------------------------------------------------- #

_omp_lib = None


def _load_omp():
    global _omp_lib
    if _omp_lib is None:
        path = os.path.join(_HERE, "_build", "liboracle_spmm.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.oracle_spmm_csr_f32.argtypes = [
                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                ctypes.c_int64, ctypes.c_int]
            lib.oracle_spmm_csr_f32.restype = None
            lib.oracle_num_threads.restype = ctypes.c_int
            lib.oracle_set_threads.argtypes = [ctypes.c_int]
            _omp_lib = lib
        else:
            _omp_lib = False
    return _omp_lib


def omp_available() -> bool:
    return bool(_load_omp())


def set_omp_threads(n: int) -> None:
    lib = _load_omp()
    if lib:
        lib.oracle_set_threads(int(n))


def omp_threads() -> int:
    lib = _load_omp()
    return int(lib.oracle_num_threads()) if lib else 1


def spmm_csr_numpy(indptr, indices, w, x: np.ndarray, mean: bool = False) -> np.ndarray:
    """Row-by-row CSR SpMM in the dtype of ``x`` (sequential in-row order)."""
    n = len(indptr) - 1
    out = np.zeros((n, x.shape[1]), dtype=x.dtype)
    for v in range(n):
        lo, hi = int(indptr[v]), int(indptr[v + 1])
        if hi == lo:
            continue
        rows = x[indices[lo:hi]]
        if w is not None:
            rows = rows * w[lo:hi, None].astype(x.dtype)
        acc = np.zeros(x.shape[1], dtype=x.dtype)
        for r in rows:                      # fixed sequential order
            acc = acc + r
        out[v] = acc / x.dtype.type(hi - lo) if mean else acc
    return out


def spmm_csr_torch(indptr, indices, w, x: torch.Tensor) -> torch.Tensor:
    """CSR SpMM on CPU: OpenMP C kernel when built, else torch.sparse_csr @ x."""
    n = len(indptr) - 1
    lib = _load_omp()
    if lib and x.dtype == torch.float32:
        x = x.contiguous()
        out = torch.empty((n, x.shape[1]), dtype=torch.float32)
        ip = np.ascontiguousarray(indptr, dtype=np.int32)
        ix = np.ascontiguousarray(indices, dtype=np.int32)
        wp = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
        lib.oracle_spmm_csr_f32(
            ip.ctypes.data, ix.ctypes.data, None if wp is None else wp.ctypes.data,
            x.data_ptr(), out.data_ptr(), n, x.shape[1], x.stride(0), out.stride(0), 0)
        return out
    vals = torch.ones(len(indices), dtype=x.dtype) if w is None else torch.as_tensor(w).to(x.dtype)
    a = torch.sparse_csr_tensor(torch.as_tensor(np.asarray(indptr, dtype=np.int64)),
                                torch.as_tensor(np.asarray(indices, dtype=np.int64)),
                                vals, size=(n, x.shape[0]))
    return a @ x


class _SpMM(torch.autograd.Function):
    """out = A_w x with backward dx = A_w^T dout (DGL GSpMM.backward; edge
    weights are constants in the reference so no SDDMM term)."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return spmm_csr_torch(graph.indptr, graph.indices, graph.weight, x)

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        return spmm_csr_torch(g.rev_indptr, g.rev_indices, g.rev_weight, dout.contiguous()), None


class OracleGraph:
    """In-edge CSR (+ out-edge CSR for backward) of one (batched) page graph."""

    def __init__(self, src, dst, num_nodes: int, weight=None):
        self.num_nodes = int(num_nodes)
        self.src = np.asarray(src, dtype=np.int64)
        self.dst = np.asarray(dst, dtype=np.int64)
        self.eweight = None if weight is None else np.asarray(weight, dtype=np.float32)
        self.indptr, self.indices, self.weight, _ = coo_to_in_csr(self.src, self.dst, num_nodes, self.eweight)
        # reversed graph: row u lists destinations of u's out-edges
        self.rev_indptr, self.rev_indices, self.rev_weight, _ = coo_to_in_csr(
            self.dst, self.src, num_nodes, self.eweight)
        self.norm = in_degree_norm(self.indptr)

    def num_edges(self) -> int:
        return len(self.src)


# --------------------------------------------------------------------------- #
# model restatement (functional, driven by a reference-format state_dict)
# --------------------------------------------------------------------------- #


def layer_forward(graph: OracleGraph, h: torch.Tensor, weight: torch.Tensor,
                  bias: Optional[torch.Tensor], ln_weight: Optional[torch.Tensor],
                  ln_bias: Optional[torch.Tensor], activation: bool,
                  use_pp: bool = False, eps: float = 1e-5) -> torch.Tensor:
    """One GcnSAGELayer.forward (models.py:46-72), dropout p=0."""
    if not use_pp:
        norm = torch.from_numpy(graph.norm).to(h.dtype)              # :74-78
        ah = _SpMM.apply(h, graph)                                    # :53-57
        h = torch.cat((h, ah * norm), dim=1)                          # :69-72
    z = torch.nn.functional.linear(h, weight, bias)                   # :63
    if ln_weight is not None:
        z = torch.nn.functional.layer_norm(z, (z.shape[1],), ln_weight, ln_bias, eps)   # :64
    if activation:
        z = torch.relu(z)                                             # :65-66
    return z


def gcnsage_forward(state: Dict[str, torch.Tensor], graph: OracleGraph, x: torch.Tensor,
                    return_hidden: bool = False):
    """GcnSAGE.forward (models.py:105-116) from a reference-format state_dict
    (keys ``layers.{i}.linear.{weight,bias}``, ``layers.{i}.lynorm.{weight,bias}``)."""
    n_layers = 1 + max(int(k.split(".")[1]) for k in state)
    h = x
    hidden = []
    for i in range(n_layers):
        last = i == n_layers - 1
        h = layer_forward(
            graph, h,
            state[f"layers.{i}.linear.weight"], state.get(f"layers.{i}.linear.bias"),
            state.get(f"layers.{i}.lynorm.weight"), state.get(f"layers.{i}.lynorm.bias"),
            activation=not last)
        hidden.append(h)
    return (h, hidden) if return_hidden else h


def init_state(in_feats: int, n_hidden: int, n_classes: int, n_layers: int,
               seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Parameters with the reference's distribution (models.py:40-44:
    U(-1/sqrt(2*in), 1/sqrt(2*in)) for W and b; LayerNorm ones/zeros).  NOT the
    reference's RNG stream -- golden fixtures carry the reference's own state_dict."""
    gen = torch.Generator().manual_seed(seed)
    dims = [in_feats] + [n_hidden] * (n_layers - 1) + [n_classes]
    state = {}
    for i in range(n_layers):
        fin, fout = dims[i], dims[i + 1]
        stdv = 1.0 / math.sqrt(2 * fin)
        state[f"layers.{i}.linear.weight"] = ((torch.rand(fout, 2 * fin, generator=gen) * 2 - 1) * stdv).to(dtype)
        state[f"layers.{i}.linear.bias"] = ((torch.rand(fout, generator=gen) * 2 - 1) * stdv).to(dtype)
        if i != n_layers - 1:
            state[f"layers.{i}.lynorm.weight"] = torch.ones(fout, dtype=dtype)
            state[f"layers.{i}.lynorm.bias"] = torch.zeros(fout, dtype=dtype)
    return state


def meansage_forward(weights: Sequence[Tuple[torch.Tensor, torch.Tensor]], graph: OracleGraph,
                     x: torch.Tensor) -> torch.Tensor:
    """MeanSAGE.forward (models.py:154-170): WeightedMeanSAGELayer = Linear(cat(h,
    mean_in-edges(w_e h[u]))) ; relu + L2-row-normalise between layers."""
    deg = np.diff(graph.indptr.astype(np.int64)).astype(np.float32)
    inv = torch.from_numpy(np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0).astype(np.float32)).to(x.dtype)[:, None]
    h = x
    for li, (w, b) in enumerate(weights):
        h_n = _SpMM.apply(h, graph) * inv                             # :146-149 fn.mean
        h = torch.nn.functional.linear(torch.cat([h, h_n], dim=1), w, b)
        if li != len(weights) - 1:
            h = torch.nn.functional.normalize(torch.relu(h))          # :166-168
    return h


# --------------------------------------------------------------------------- #
# one optimisation step (model_train.py:168-171, :320-332)
# --------------------------------------------------------------------------- #


class OracleTrainer:
    """CE(weight) -> backward -> torch.optim.Adam(lr, weight_decay) on leaf copies
    of a state_dict.  Adam here is torch's own (L2-coupled decay) = the reference's."""

    def __init__(self, state: Dict[str, torch.Tensor], lr: float = 0.01, weight_decay: float = 5e-4,
                 class_weights: Optional[torch.Tensor] = None):
        self.state = {k: v.clone().requires_grad_(True) for k, v in state.items()}
        self.opt = torch.optim.Adam(list(self.state.values()), lr=lr, weight_decay=weight_decay)
        self.loss_fn = torch.nn.CrossEntropyLoss(weight=class_weights)

    def step(self, graph: OracleGraph, x: torch.Tensor, labels: torch.Tensor):
        logits = gcnsage_forward(self.state, graph, x)
        loss = self.loss_fn(logits, labels.long())                    # model_train.py:327
        self.opt.zero_grad()
        loss.backward()                                               # :331
        self.opt.step()                                               # :332
        return float(loss.detach()), logits.detach()

    def grads(self) -> Dict[str, torch.Tensor]:
        return {k: v.grad.detach().clone() for k, v in self.state.items()}


# --------------------------------------------------------------------------- #
# independent fp64 dense-adjacency formulation (cross-check of the restatement)
# --------------------------------------------------------------------------- #


def dense_reference_forward(state: Dict[str, np.ndarray], src, dst, weight, num_nodes: int,
                            x: np.ndarray) -> np.ndarray:
    """ah = (A o W) @ X with A[v,u] += w per edge u->v, all in float64 numpy."""
    a = np.zeros((num_nodes, num_nodes), dtype=np.float64)
    w = np.ones(len(src)) if weight is None else np.asarray(weight, dtype=np.float64)
    np.add.at(a, (np.asarray(dst), np.asarray(src)), w)
    deg = np.bincount(np.asarray(dst), minlength=num_nodes).astype(np.float64)
    norm = np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0)[:, None]
    n_layers = 1 + max(int(k.split(".")[1]) for k in state)
    h = x.astype(np.float64)
    for i in range(n_layers):
        wgt = np.asarray(state[f"layers.{i}.linear.weight"], dtype=np.float64)
        b = np.asarray(state[f"layers.{i}.linear.bias"], dtype=np.float64)
        z = np.concatenate([h, (a @ h) * norm], axis=1) @ wgt.T + b
        if f"layers.{i}.lynorm.weight" in state:
            g = np.asarray(state[f"layers.{i}.lynorm.weight"], dtype=np.float64)
            be = np.asarray(state[f"layers.{i}.lynorm.bias"], dtype=np.float64)
            mu = z.mean(1, keepdims=True)
            var = z.var(1, keepdims=True)
            z = (z - mu) / np.sqrt(var + 1e-5) * g + be
        if i != n_layers - 1:
            z = np.maximum(z, 0.0)
        h = z
    return h


# --------------------------------------------------------------------------- #
# shape helpers (components/features/utils.py:71-101)
# --------------------------------------------------------------------------- #

FEATURE_WIDTHS = {"BBOX": 13, "REPR": 50, "SPACY": 300, "SCIBERT": 768}


def get_in_feats(features: Sequence[str], padding: bool = False) -> int:
    if padding:                                                        # utils.py:80-83
        features = ["BBOX", "REPR", "SCIBERT"]
    return sum(FEATURE_WIDTHS[f] for f in features)


def calculate_hidden(input_dim: int, classes_no: int, params_no: int, layer_no: int) -> float:
    """Largest root of (L-1) h^2 + (C+F0) h - P = 0  (utils.py:90-101)."""
    a = layer_no - 1
    b = classes_no + input_dim
    delta = b * b + 4 * a * params_no
    return max((-b - math.sqrt(delta)) / (2 * a), (-b + math.sqrt(delta)) / (2 * a))
