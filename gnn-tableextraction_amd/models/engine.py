"""One optimisation step of the node classifier on the HIP path, single- or multi-GPU.

Restates the reference's batch loop body (src/models/model_train.py:320-332:
``logits = model(g); loss = CE(logits, labels.long()); zero_grad; backward; optimizer.step()``)
over ONE flat fp32 parameter buffer and ONE flat gradient buffer:
  * the model's parameters are views into ``flat_param``; their ``.grad`` are views into
    ``flat_grad`` (autograd accumulates in place), so the optimiser is one fused Adam launch
    (gte_adam_step: torch.optim.Adam semantics, L2-coupled weight decay) and
  * data parallelism is ONE RCCL all-reduce of ``flat_grad`` per step (page graphs never share
    edges, so there is no other exchange).  Each rank's loss is a mean over ITS nodes; scaling it
    by n_local / n_global before backward makes the summed gradient equal to the single-GPU
    gradient of the mean over all nodes.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np
import torch

from .. import ops


class TrainStep:
    def __init__(self, model: torch.nn.Module, lr: float = 0.01, weight_decay: float = 5e-4,
                 class_weights: Optional[torch.Tensor] = None, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, distributed: bool = False):
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.class_weights = class_weights
        self.distributed = distributed
        self.group = process_group
        params = [p for p in model.parameters() if p.requires_grad]
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p)
            p.grad = self.flat_grad[off:off + n].view_as(p)
            off += n
        self.t = 0
        if distributed:
            import torch.distributed as dist
            dist.broadcast(self.flat_param, src=0, group=process_group)   # same initial weights everywhere

    def _dp_scale(self, n_local: int, n_global: Optional[int], loss_scale: Optional[float]) -> float:
        """Factor on the local loss so that the SUM of the ranks' gradients is the single-GPU gradient.
        Unweighted CE is a mean over nodes: n_local / n_global.  With class weights the local loss is
        sum(w nll) / sum_local(w) (nn.CrossEntropyLoss(weight), model_train.py:171) and the factor is
        sum_local(w) / sum_global(w): the caller, who knows every rank's labels from the shared plan, passes it as
        ``loss_scale`` -- a node-count ratio would silently optimise a different objective."""
        if loss_scale is not None:
            return float(loss_scale)
        if not (self.distributed and n_global):
            return 1.0
        if self.class_weights is not None:
            raise ValueError("data-parallel step with class weights needs loss_scale = sum_local(w) / sum_global(w) "
                             "(distributed.step_weight_sums); n_local / n_global is only right for unweighted CE")
        return float(n_local) / float(n_global)

    def step(self, g, labels: torch.Tensor, n_global: Optional[int] = None,
             loss_scale: Optional[float] = None) -> torch.Tensor:
        """Forward, loss, backward, (all-reduce), Adam.  Returns the device vector
        [loss (local mean), sum of class weights, #correct] without synchronising."""
        self.model.train()
        self.flat_grad.zero_()
        logits = self.model(g)
        loss, out3 = self._loss(logits, labels)
        scale = self._dp_scale(labels.shape[0], n_global, loss_scale)
        if scale != 1.0:
            loss = loss * scale
        loss.backward()
        if self.distributed:
            import torch.distributed as dist
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        self.t += 1
        self._optimizer_step()
        return out3

    # the two arithmetic pieces of the step; HIP here (tests of the DP logic override them on CPU)
    def _loss(self, logits, labels):
        return ops.cross_entropy(logits, labels, self.class_weights)

    def _optimizer_step(self) -> None:
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.t, self.lr,
                      self.betas[0], self.betas[1], self.eps, self.weight_decay)

    # checkpoint-compatible with torch.optim.Adam's state_dict layout is handled by model_train.py
    def lr_scale(self, factor: float) -> None:
        self.lr *= factor


# ======================================================================================================
# Hand-scheduled step for GcnSAGE: no autograd, no per-step allocation, HIP-graph capturable
# ======================================================================================================
from .. import _lib                                                     # noqa: E402
from ..components.graphs.models import GcnSAGE, _is_relu               # noqa: E402
import torch.nn as nn                                                  # noqa: E402


def _c16(x: int) -> int:
    return -(-int(x) // 16) * 16


class FusedGcnSageStep(TrainStep):
    """The same optimisation step as :class:`TrainStep` for a :class:`GcnSAGE` model, scheduled by hand:

    * forward and backward are explicit sequences of C-ABI calls (aggregation, split-weight GEMM,
      LayerNorm/ReLU, CE, LayerNorm backward, dW / dX GEMMs, transpose aggregation) -- no autograd
      graph, no Python allocation per step (buffers are cached per node count);
    * parameter gradients are written by the kernels DIRECTLY into their slices of ``flat_grad`` (every
      producer overwrites, so the buffer is never zeroed and never accumulated into);
    * nothing in a step synchronises or allocates, so a step on a resident batch can be captured
      into a HIP graph (:meth:`capture`) and replayed with one launch -- the page-batch regime is a few
      dozen 10-300 us kernels, where per-launch host time would otherwise bound the step.
    Requires the configuration every shipped reference run uses: ReLU (or no) activation, dropout 0,
    use_pp False; anything else goes through the autograd path (:class:`TrainStep`).
    """

    def __init__(self, model: GcnSAGE, **kw):
        if not isinstance(model, GcnSAGE):
            raise TypeError("FusedGcnSageStep needs a GcnSAGE model")
        for layer in model.layers:
            if layer.use_pp or (layer.dropout and layer.dropout.p > 0) or not (layer.activation is None or _is_relu(layer.activation)):
                raise ValueError("FusedGcnSageStep supports ReLU/None activations, dropout 0, use_pp False")
        if model.dropout.p > 0:
            raise ValueError("FusedGcnSageStep supports dropout 0 only")
        super().__init__(model, **kw)
        self.lib = _lib.load()
        # slices of the flat gradient, in model.parameters() order
        self._gslice = {}
        off = 0
        for p in (q for q in model.parameters() if q.requires_grad):
            self._gslice[id(p)] = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        # data-parallel overlap (GTE_DP_OVERLAP=1): the flat gradient is [layer 0 | layers 1..]; the upper slice is all-reduced
        # while layer 0's backward (the longest: its dW GEMM alone is a quarter of the step) still runs.  Off by default: with
        # ONE rank (RCCL process group, trivial collective) the split step costs 0.804 ms against 0.760 ms for one all-reduce
        # behind one graph (0.748 ms without data parallelism) -- cutting the graph and the second collective cost ~45 us,
        # about what hiding a 0.5 MB all-reduce can win back on 8 GPUs.  To be re-measured on a multi-GPU node.
        self._n0 = sum(p.numel() for p in model.layers[0].parameters() if p.requires_grad)
        first_upper = next((p for p in model.layers[1].parameters() if p.requires_grad), None) if len(model.layers) > 1 else None
        self._dp_split = (os.environ.get("GTE_DP_OVERLAP", "0") == "1" and first_upper is not None
                          and self._gslice[id(first_upper)].storage_offset() == self._n0)
        self._bufs = {}
        self._graph_bufs = {}
        self._private_key = None
        self._graphs = {}
        self._graph_owner = {}
        # transform-then-aggregate where a layer narrows + q-form backward (see _transform_first / _qform); "0" keeps
        # the reference's aggregate-then-transform order everywhere (same math, different summation order)
        self.transform_first = os.environ.get("GTE_TRANSFORM_FIRST", "1") == "1"
        self.tail_split = os.environ.get("GTE_TAIL_SPLIT", "1") == "1"
        # single-GPU steps: the Adam update runs inside the gradient-fold launch (gte_fold_defer_flush_adam)
        self.fuse_adam = os.environ.get("GTE_FUSE_ADAM", "1") == "1"
        self._fuse_adam_req, self._adam_fused = False, False
        self.adam_fused_steps = 0                     # steps whose optimiser update ran inside the fold launch
        self.fused_head = os.environ.get("GTE_FUSED_HEAD", "1") == "1"
        self.fuse_head_gemm = os.environ.get("GTE_FUSE_HEAD_GEMM", "1") == "1"     # (general plans: the fused head on the GEMM output path)
        # dX of a planes layer with the LayerNorm(+ReLU) backward of the planes layer below as its epilogue (gte_gemm_p3_nt_ln_bwd)
        self.fuse_ln_dx = os.environ.get("GTE_FUSE_LN_DX", "1") == "1"
        # ... and of the last hidden layer inside the output layer's backward (gte_sage_narrow_bwd_ln_p3)
        self.fuse_ln_narrow = os.environ.get("GTE_FUSE_LN_NARROW", "1") == "1"
        # dX of layer 1 with the whole backward of a short-input layer 0 as its epilogue (gte_gemm_p3_nt_smallk_bwd)
        self.fuse_smallk_dx = os.environ.get("GTE_FUSE_SMALLK_DX", "1") == "1"
        self._smallk_done = False
        self._ln_p3_done = None
        # LayerNorm(+ReLU) forward of the last hidden layer inside the output layer's forward kernel (gte_sage_narrow_fwd_ln):
        # one launch and one pass over [n, hidden] less
        self.fuse_ln_fwd = os.environ.get("GTE_FUSE_LN_FWD", "1") == "1"
        self._head_scale = None
        self._tail_ws = None
        # planes path (GTE_PLANES=0 disables): in the split-bf16 GEMM mode the operands of the transform GEMMs are written as P3
        # images (three bf16 planes, csrc/p3.h) by their producers and multiplied by the planes GEMMs (csrc/gemm_p3.hip)
        self.use_planes = os.environ.get("GTE_PLANES", "1") == "1"
        # ... for EVERY hidden width up to 1024 and any input width through the one-call plan (padded rows, masked LayerNorm
        # kernels, aggregate-first input layer, output layer on the planes GEMMs); GTE_PLANES_GENERAL=0: the tuned range only
        self.general_planes = os.environ.get("GTE_PLANES_GENERAL", "1") == "1"
        # input layer on the CACHED mean aggregate of the input (graph.ResidentPages.build_agg_image: page-local, constant over a
        # run): z = [x | ahn] W^T from two resident images behind the batch's row map, no aggregation / q / feature copy in the
        # step for layer 0 (GTE_LAYER_CACHED).  Costs a second resident image; GTE_CACHE_AGG=0 turns it off
        self.cache_input_agg = os.environ.get("GTE_CACHE_AGG", "1") == "1"
        self._wimg = {}                               # layer index -> (forward image, backward image or None)
        # weight images in the block-major layout (ops.P3): a K block of the weights is ONE contiguous run, whole cache lines for
        # every NT planes GEMM, and the block-major-weights kernel (gemm_p3_nt_sq_kernel) loads its fragments straight into
        # registers.  GTE_WIMG_BLOCK_MAJOR=0: row-major images (the layout up to round 5; A/B measurements, tests)
        self.block_major_weights = os.environ.get("GTE_WIMG_BLOCK_MAJOR", "1") == "1"
        # one-call step: the fold + Adam launch also writes the weight images of the updated parameters
        # (gte_fold_defer_flush_adam_images) and the next step's forward skips their conversion launch.  _wimg_sig = the version
        # counters of the parameters the images were made from (None: stale).  In-place writes through torch (load_state_dict,
        # p.copy_, flat_param.copy_) move the counters; raw writes through ``p.data`` do not: invalidate_weight_images() then.
        self.wimg_in_fold = os.environ.get("GTE_WIMG_IN_FOLD", "1") == "1"
        self._wimg_sig = None
        # the whole step through ONE C entry point (gte_gcnsage_step) when the configuration allows (GTE_C_STEP=0: call by call)
        self.use_c_step = os.environ.get("GTE_C_STEP", "1") == "1"
        # called (once per step, no arguments) right before the LAST big kernel of a step is launched -- layer 0's dW GEMM,
        # MFMA-bound, ~a quarter of the step.  The train loop hangs the assembly of the NEXT batch here (models/loop.py):
        # it then runs on the side stream under that GEMM, and the batch is still in the Infinity Cache when the next
        # step's first GEMM reads it.
        self.before_last_gemm = None
        # measurement hook of the one-call step (bench.py): a list of 2 x n_hidden torch.cuda.Event(enable_timing=True), recorded
        # around the forward transform GEMM of every hidden layer of the NEXT steps (None: off)
        self.fwd_events = None
        self._fwd_ev_arr = None

    # -- buffers -------------------------------------------------------------------------------------
    def _alloc(self, cap: int, f0: int):
        dev = self.flat_param.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        layers = self.model.layers
        dims = [f0] + [l.out_feats for l in layers]
        tf = [self._transform_first(l, dims[i]) for i, l in enumerate(layers)]
        qf = [self._qform(i, l, dims[i]) for i, l in enumerate(layers)]
        b = {"cap": cap,
             # aggregated input [cap, fin] of an aggregate-first layer; a q-form layer reuses it for q in the backward
             "ahn": [None if tf[i] else new(cap, dims[i]) for i in range(len(layers))],
             # transform-first layer: t = [x W_s^T + b | x W_n^T]; z is accumulated into the left half, q reuses the right
             "t": [new(cap, 2 * dims[i + 1]) if tf[i] else None for i in range(len(layers))],
             "z": [new(cap, dims[i + 1]) if (isinstance(l.lynorm, nn.LayerNorm) and not tf[i]) else None
                   for i, l in enumerate(layers)],
             "stats": [new(2 * cap) if isinstance(l.lynorm, nn.LayerNorm) else None for l in layers],
             "y": [new(cap, dims[i + 1]) for i in range(len(layers))],
             "dy": [new(cap, dims[i + 1]) for i in range(len(layers))],      # grad w.r.t. layer output (dz in place)
             "dahn": new(cap, max([dims[i] for i in range(1, len(layers)) if not qf[i] and not self._narrow(layers[i], dims[i])]
                                  + [1])),
             "tn": new(cap, dims[-1]), "q": new(cap, dims[-1]),       # narrow (class-count-wide) output layer
             "out3": new(3)}
        lib = self.lib
        # planes layers: P3 images of the layer input (layer 0: only when the batch does not bring one), of dz and of q, and
        # the split-K workspace of the dW planes GEMM
        pl = [self._planes_layer(i, l, dims[i]) for i, l in enumerate(layers)]
        b["pl"] = pl
        b["hp"] = [ops.P3.empty(cap, dims[i], dev) if pl[i] else None for i in range(len(layers))]
        b["dzp"] = [ops.P3.empty(cap, dims[i + 1], dev) if pl[i] else None for i in range(len(layers))]
        b["qp"] = [ops.P3.empty(cap, dims[i + 1], dev) if pl[i] else None for i in range(len(layers))]
        b["ws_p3"] = [torch.empty(int(lib.gte_gemm_p3_tn_workspace_bytes(dims[i + 1], 2 * dims[i], dims[i], cap)), dtype=torch.uint8,
                                  device=dev) if pl[i] else None for i in range(len(layers))]
        for i in range(len(layers)):
            if pl[i] and b["t"][i] is None:
                b["t"][i] = new(cap, 2 * dims[i + 1])
        # every workspace requirement grows with the node count, so the capacity's requirement covers any n <= cap
        ws = max([lib.gte_weighted_ce_workspace_bytes(cap)] +
                 [lib.gte_ln_relu_bwd_workspace_bytes(cap, d) for d in dims[1:]] +
                 [lib.gte_gemm_workspace_bytes(dims[i + 1], dims[i], cap) for i in range(len(layers))] +
                 [lib.gte_sage_narrow_bwd_workspace_bytes(cap, min(dims[-2], 256), min(dims[-1], 16))])
        b["ws"] = torch.empty(int(ws), dtype=torch.uint8, device=dev)
        # the backward defers its partial-sum folds to one launch (gte_fold_defer_*): every producer keeps its partials
        # in a workspace of its own until the flush
        b["ws_ln"] = [torch.empty(int(max(lib.gte_ln_relu_bwd_workspace_bytes(cap, dims[i + 1]),
                                          lib.gte_gemm_p3_nt_ln_bwd_workspace_bytes(cap, dims[i + 1]),
                                          lib.gte_sage_narrow_bwd_ln_workspace_bytes(cap, min(dims[i + 1], 256)))),
                                  dtype=torch.uint8, device=dev) for i in range(len(layers))]
        b["ce_part"] = torch.empty(int(lib.gte_head_agg_ce_workspace_bytes(cap)), dtype=torch.uint8, device=dev)
        b["ws_nar"] = torch.empty(int(lib.gte_sage_narrow_bwd_workspace_bytes(cap, min(dims[-2], 256), min(dims[-1], 16))),
                                  dtype=torch.uint8, device=dev)
        # one private workspace per layer for the dW GEMMs: they run on the side stream, several at once
        b["ws_dw"] = [torch.empty(int(self._ws_dw_bytes(i, dims, cap)), dtype=torch.uint8, device=dev) for i in range(len(layers))]
        return b

    def _ws_dw_bytes(self, i: int, dims, cap: int) -> int:
        """Workspace of layer i's weight-gradient launch: split-K slabs of the dW GEMM; for the input layer additionally the
        partials of the one-pass short-input backward -- only where that path exists (k1 + k2 <= 28: at F0 = 831 the size
        functions, which do not check support, would ask for 764 + 382 MB per buffer set)."""
        lib = self.lib
        need = [lib.gte_sage_linear_dw_workspace_bytes(dims[i + 1], dims[i], dims[i], cap),
                lib.gte_sage_qform_dw_workspace_bytes(dims[i + 1], dims[i], cap)]
        if i == 0 and lib.gte_sage_smallk_bwd_supported(2 * dims[0], dims[1]):
            need.append(lib.gte_sage_smallk_bwd_workspace_bytes(cap, 2 * dims[0], dims[1]))
        if i == 0 and len(dims) > 2 and lib.gte_gemm_p3_nt_smallk_bwd_supported(2 * dims[0], dims[1]):
            need.append(lib.gte_gemm_p3_nt_smallk_bwd_workspace_bytes(cap, 2 * dims[0], dims[1]))
        return max(need)

    # -- buffers of the one-call plan on general widths (padded rows) ----------------------------------
    def _alloc_gen(self, cap: int, f0: int, kinds, out_gemm: bool):
        """Buffer set of a one-call plan whose layers are not all in the tuned range (128 <= hidden <= 256, hidden % 16 == 0):
        the fp32 row buffers of a hidden layer are PADDED to ld = hidden rounded up to 16 floats (include/gte.h, gte_step_layer.ldf),
        zero-initialised once (the kernels keep the padding zero); images are padded to 16-column blocks by construction."""
        dev, lib = self.flat_param.device, self.lib
        layers = list(self.model.layers)
        dims = [f0] + [l.out_feats for l in layers]
        nh = len(layers) - 1
        ld = [_c16(d) for d in dims[1:nh + 1]]
        z32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        u8 = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device=dev)

        def img(rows, cols):
            t = ops.P3.empty(rows, cols, dev)
            t.data.zero_()
            return t
        b = {"cap": cap, "gen": True, "kinds": tuple(kinds), "out_gemm": bool(out_gemm), "ld": ld, "pl": [k == 0 for k in kinds]}
        b["t"] = [z32(cap, 2 * ld[i]) if kinds[i] == 0 else z32(cap, ld[i]) for i in range(nh)]       # [t_self | t_neigh], or z
        b["y"] = [z32(cap, ld[i]) for i in range(nh)]
        b["dy"] = [z32(cap, ld[i]) for i in range(nh)]
        b["stats"] = [z32(2 * cap) for _ in range(nh)]
        b["ahn"] = [z32(cap, dims[0]) if kinds[i] == 1 else None for i in range(nh)]
        b["ahnp"] = [img(cap, dims[0]) if kinds[i] == 2 else None for i in range(nh)]
        b["hp"] = [img(cap, dims[i]) if kinds[i] not in (1, 3) else None for i in range(nh)]      # (3: both operands are resident)
        b["dzp"] = [img(cap, dims[i + 1]) if kinds[i] != 1 else None for i in range(nh)]
        b["qp"] = [img(cap, dims[i + 1]) if kinds[i] == 0 else None for i in range(nh)]
        b["ws_dw"] = [u8(lib.gte_gemm_p3_tn_workspace_bytes(dims[i + 1], 2 * dims[i], dims[i], cap) if kinds[i] != 1
                         else self._ws_dw_bytes(i, dims, cap)) for i in range(nh)]
        b["ws_ln"] = [u8(max(lib.gte_ln_relu_bwd_workspace_bytes(cap, dims[i + 1]),
                             lib.gte_gemm_p3_nt_ln_bwd_workspace_bytes(cap, dims[i + 1]) if dims[i + 1] <= 256 else 0,
                             lib.gte_sage_narrow_bwd_ln_workspace_bytes(cap, min(_c16(dims[i + 1]), 256)))) for i in range(nh)]
        C = dims[-1]
        b["out3"] = z32(3)
        if out_gemm:
            b["hp_out"] = img(cap, dims[-2])                       # image of the last hidden layer's output
            b["t_out"], b["dlq"] = z32(cap, 32), z32(cap, 32)      # [logits .. | t_neigh ..], [dl .. | q ..]
            b["dlqp"] = img(cap + 8, 32)
            b["ws_out"] = u8(lib.gte_gemm_p3_tn_workspace_bytes(C, 2 * dims[-2], dims[-2], cap))
            b["ws_ce"] = u8(max(lib.gte_weighted_ce_workspace_bytes(cap), lib.gte_head_agg_ce_workspace_bytes(cap)))
            b["ws_cs"] = u8(max(lib.gte_colsum_workspace_bytes(cap, C), lib.gte_head_dlq_finish_workspace_bytes(cap)))
        else:
            b["logits"], b["tn"], b["q"], b["dl"] = z32(cap, C), z32(cap, C), z32(cap, C), z32(cap, C)
            b["ce_part"] = u8(lib.gte_head_agg_ce_workspace_bytes(cap))
            b["ws_nar"] = u8(lib.gte_sage_narrow_bwd_workspace_bytes(cap, min(_c16(dims[-2]), 256), min(C, 16)))
        return b

    def _buffers_gen(self, n: int, f0: int, kinds, out_gemm: bool):
        """The (capacity-sized, shared or captured-batch-private) buffer set of a general plan; see _buffers."""
        key = ("gen", f0, tuple(kinds), bool(out_gemm))
        if self._private_key is not None:
            store, k2 = self._graph_bufs, (self._private_key, key)
            full = store.get(k2)
            if full is None:
                full = store[k2] = self._alloc_gen(n, f0, kinds, out_gemm)
            if full["cap"] < n:
                raise RuntimeError(f"captured batch buffers hold {full['cap']} nodes; asked for {n} (a captured batch must not change)")
            return full
        full = self._bufs.get(key)
        if full is None or full["cap"] < n:
            cap = max(-(-int(n * 1.125) // 4096) * 4096, getattr(self, "_reserved", {}).get(f0, 0))
            self._bufs.pop(key, None)
            full = self._bufs[key] = self._alloc_gen(cap, f0, kinds, out_gemm)
        return full

    def _buffers(self, n: int, f0: int, private=None):
        """Row views [0:n] of buffers allocated for a CAPACITY, not for n: in the real loop every batch has a
        different node count, and per-count buffers would grow without bound (~300 MB per new count at F0=831).
        The shared set grows geometrically to the largest batch seen.  A captured HIP graph bakes pointers in,
        so each captured batch owns a private exact-size set (``private`` = its key) that is never reallocated."""
        if private is not None:
            full = self._graph_bufs.get(private)
            if full is None:
                full = self._graph_bufs[private] = self._alloc(n, f0)
                full["f0"] = f0
            if full["cap"] < n or full["f0"] != f0:
                raise RuntimeError(f"captured batch buffers hold {full['cap']} nodes x {full['f0']} features; "
                                   f"asked for {n} x {f0} (a captured batch must not change)")
        else:
            key = (f0, self._planes_on())              # the layer plan (which layers take P3 operands) depends on the GEMM mode
            full = self._bufs.get(key)
            if full is None or full["cap"] < n:
                cap = -(-int(n * 1.125) // 4096) * 4096
                self._bufs = {k: v for k, v in self._bufs.items() if k[0] != f0}     # one set per input width alive
                full = self._bufs[key] = self._alloc(cap, f0)
        v = lambda t: None if t is None else t[:n]
        return {"ahn": [v(t) for t in full["ahn"]], "t": [v(t) for t in full["t"]], "z": [v(t) for t in full["z"]],
                "stats": [None if t is None else t[:2 * n] for t in full["stats"]], "y": [v(t) for t in full["y"]],
                "dy": [v(t) for t in full["dy"]], "dahn": v(full["dahn"]), "tn": v(full["tn"]), "q": v(full["q"]),
                "out3": full["out3"], "ws": full["ws"], "ws_ln": full["ws_ln"], "ws_nar": full["ws_nar"],
                "ce_part": full["ce_part"],
                "ws_dw": full["ws_dw"], "pl": full["pl"], "ws_p3": full["ws_p3"], "_full": full,
                "hp": [None if t is None else t.view_rows(n) for t in full["hp"]],
                "dzp": [None if t is None else t.view_rows(n) for t in full["dzp"]],
                "qp": [None if t is None else t.view_rows(n) for t in full["qp"]]}

    def reserve(self, n_nodes: int, f0: int, cached: bool = False) -> None:
        """Size the shared per-batch buffers for batches of up to ``n_nodes`` nodes now (the train loop knows the largest batch
        its page table can produce): no reallocation -- a device synchronisation plus ~12 KB per node of new buffers -- later,
        in the middle of an epoch."""
        cap = -(-int(n_nodes) // 4096) * 4096
        if not hasattr(self, "_reserved"):
            self._reserved = {}
        self._reserved[f0] = max(self._reserved.get(f0, 0), cap)
        kinds = self._plan_kinds(f0, 0, cached)     # (cached: the batches bring the aggregate image -> the general plan's buffer set)
        if kinds is not None and self._plan_mode(kinds, f0)[0]:
            return                      # a general plan allocates its own (padded) set at this capacity on first use
        key = (f0, self._planes_on())
        full = self._bufs.get(key)
        if n_nodes > 0 and (full is None or full["cap"] < n_nodes):
            self._bufs = {k: v for k, v in self._bufs.items() if k[0] != f0}    # drop the old set first: both need not be alive
            self._bufs[key] = self._alloc(cap, f0)

    def _narrow(self, layer, fin: int) -> bool:
        return (not isinstance(layer.lynorm, nn.LayerNorm) and layer.activation is None and layer.linear.bias is not None
                and bool(self.lib.gte_sage_narrow_supported(fin, layer.out_feats)))

    def _ln_rows_below(self, i: int, layers, fin: int, b) -> bool:
        """The output layer's backward runs the LayerNorm(+ReLU) backward of the PLANES layer below in the row form
        (gte_sage_narrow_bwd_ln_p3: dz as fp32 + image)."""
        return (self.fuse_ln_narrow and i > 0 and i == len(layers) - 1 and bool(b["pl"][i - 1])
                and self._narrow(layers[i], fin) and fin % 16 == 0 and bool(self.lib.gte_head_supported(fin, layers[i].out_feats)))

    def _narrow_padded(self, layer, fin: int) -> bool:
        """The narrow output kernels on PADDED hidden rows (gte_sage_narrow_fwd_pad / gte_sage_narrow_bwd_ln_p3_pad): a hidden width
        that is not a multiple of 8 -- 100 / 139 / 149 / 157 / 206 / 218 of the reference's scaled runs -- up to 256 after padding
        to 16; needs the fused head and the LayerNorm backward inside the output layer's backward (the padded form has no other)."""
        return (self.general_planes and self.fused_head and self.fuse_ln_narrow and fin % 8 != 0
                and bool(self.lib.gte_sage_narrow_pad_supported(fin, _c16(fin), layer.out_feats)))

    def _fused_head(self, i: int, layer, fin: int) -> bool:
        """Output layer + weighted CE as gte_head_agg_ce / gte_sage_narrow_bwd_ce (the last, narrow layer only)."""
        return (self.fused_head and i == len(self.model.layers) - 1 and self._narrow(layer, fin)
                and bool(self.lib.gte_head_supported(fin, layer.out_feats)))

    def _transform_first(self, layer, fin: int) -> bool:
        """Forward as  z = h W_s^T + b + mean-aggregate(h W_n^T)  when the layer narrows (831 -> 256): the aggregation
        moves out_feats columns instead of fin (gte_sage_transform_fwd)."""
        return (self.transform_first and isinstance(layer.lynorm, nn.LayerNorm) and layer.linear.bias is not None
                and not self._narrow(layer, fin) and fin > layer.out_feats)

    def _qform(self, i: int, layer, fin: int) -> bool:
        """Backward through q = A_w^T(norm * dz): dW = [dz^T h | q^T h], dh = dz W_s + q W_n.  Always for a transform-first
        layer (nothing else was saved); for an inner layer when q is not wider than the classic dahn."""
        if self._narrow(layer, fin):
            return False
        return self._transform_first(layer, fin) or (self.transform_first and i > 0 and layer.out_feats <= fin)

    # -- planes path ---------------------------------------------------------------------------------
    def _planes_on(self) -> bool:
        return self.use_planes and self.transform_first and ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16

    def _planes_layer(self, i: int, layer, fin: int, n: int = 0) -> bool:
        """Layer i runs  t = h [W_s ; W_n]^T (planes GEMM),  z = t_self + b + mean-aggregate(t_neigh), LayerNorm, ReLU  with P3
        operands: a LayerNorm layer that does not widen (the aggregation then moves out_feats <= fin columns), 16-aligned
        widths the fused aggregation + LayerNorm kernel and the P3 LayerNorm backward cover."""
        fout = layer.out_feats
        return (self._planes_on() and isinstance(layer.lynorm, nn.LayerNorm) and layer.linear.bias is not None
                and not self._narrow(layer, fin) and fin >= fout and fin >= 16 and fout % 16 == 0 and 128 <= fout <= 256
                and (i == 0 or fin % 16 == 0)              # inner layers get their input image from a producer epilogue
                and bool(self.lib.gte_spmm_csr_accumulate_ln_supported(fout))
                and not (n and ops.use_tiled(n, fout, None, fused_ln=True)))

    def wants_p3_features(self, f0: int) -> bool:
        """True when layer 0 takes its input as a P3 image: the train loop then keeps the resident features as images and
        assembles batches of image rows (graph.ResidentPages.enable_p3)."""
        L = self.model.layers[0]
        return self._layer_kind(0, L, f0) == 0

    def _param_sig(self):
        return (self.flat_param._version,) + tuple(p._version for p in self.model.parameters())

    def invalidate_weight_images(self) -> None:
        """The parameters were changed behind torch's version counters (a raw ``p.data`` write, a foreign kernel): the next
        forward converts the weight images again."""
        self._wimg_sig = None

    def _weight_images(self, dims, launch=True):
        """P3 images of the planes layers' weights: forward [W_s rows ; W_n rows] x fin, backward (dX) [fin rows] x [W_s^T | W_n^T].
        ONE launch in front of every forward (the parameters change every step; the launch is part of a captured step)."""
        layers = self.model.layers
        descs = []
        for i, L in enumerate(layers):
            fin, fout = dims[i], L.out_feats
            if not self._planes_layer(i, L, fin):
                continue
            img = self._wimg.get(i)
            if img is None:
                # (weight images are BLOCK-MAJOR: the B operand of every NT planes GEMM -- ops.P3)
                fwd = ops.P3.empty(2 * fout, fin, self.flat_param.device, block_major=self.block_major_weights)
                fwd.data.zero_()
                bwd = None
                if i > 0:
                    bwd = ops.P3.empty(fin, 2 * fout, self.flat_param.device, block_major=self.block_major_weights)
                    bwd.data.zero_()
                img = self._wimg[i] = (fwd, bwd)
            fwd, bwd = img
            W = L.linear.weight
            wp, ld = W.data_ptr(), W.stride(0)
            descs.append(_lib.P3Desc(wp, ld, fout, fin, 0, fwd.at(0, 0), fwd.ldp))
            descs.append(_lib.P3Desc(wp + 4 * fin, ld, fout, fin, 0, fwd.at(fout, 0), fwd.ldp))
            if bwd is not None:
                descs.append(_lib.P3Desc(wp, ld, fin, fout, 1, bwd.at(0, 0), bwd.ldp))
                descs.append(_lib.P3Desc(wp + 4 * fin, ld, fin, fout, 1, bwd.at(0, fout // 16), bwd.ldp))
        self._wimg_descs = descs
        if launch:
            st = _lib.current_stream()
            for k in range(0, len(descs), 16):
                chunk = descs[k:k + 16]
                arr = (_lib.P3Desc * len(chunk))(*chunk)
                _lib.check(self.lib.gte_p3_from_f32_batch(ctypes.addressof(arr), len(chunk), st), "gte_p3_from_f32_batch")

    # -- the whole step as one host call (gte_gcnsage_step) ---------------------------------------------
    def _smallk_bwd(self, i: int, layer, fin: int) -> bool:
        """Layer 0 with a short input (BBOX features) and LayerNorm: forward in one pass (gte_sage_linear_fwd_fuses_ln) and the
        whole backward in one pass (gte_sage_smallk_bwd); z is then never saved."""
        return (i == 0 and isinstance(layer.lynorm, nn.LayerNorm) and layer.linear.bias is not None
                and not self._transform_first(layer, fin) and not self._planes_layer(i, layer, fin, 1 << 20)
                and bool(self.lib.gte_sage_linear_fwd_fuses_ln(2 * fin, layer.out_feats))
                and bool(self.lib.gte_sage_smallk_bwd_supported(2 * fin, layer.out_feats)))

    def _cached_layer0(self, L, fin: int) -> bool:
        """Layer 0 can run on the cached mean aggregate of the input (GTE_LAYER_CACHED) when the batch brings it: any LayerNorm
        layer on the planes path (the tuned one-pass kernels of the 13-feature input at aligned hidden widths keep their path)."""
        # (hidden widths below 128 keep the transform-first order: their input GEMM is bound by the operand stream, and the cached
        # form reads two input images where transform-first reads one -- measured (831, 96) 62 against 71 M nodes/s, (781, 100) 58
        # against 66; from 139 columns up the cached form wins 4 - 6 %: profiles/r05/cache_agg_ab.txt)
        return (self.cache_input_agg and self.general_planes and self._planes_on() and isinstance(L.lynorm, nn.LayerNorm)
                and L.linear.bias is not None and (L.activation is None or _is_relu(L.activation))
                and 128 <= L.out_feats <= 1024)

    def wants_agg_image(self, f0: int) -> bool:
        """True when the train loop should keep the image of the input's mean aggregate next to the feature image
        (graph.ResidentPages.enable_p3(agg=True))."""
        L = self.model.layers[0]
        return len(self.model.layers) >= 2 and self._cached_layer0(L, f0) and self._layer_kind(0, L, f0) in (0, 2)

    def wants_resident_images(self, f0: int) -> bool:
        """True when the train loop should keep the resident features as images (and hand out row-map batches): layer 0 takes its
        input as an image (wants_p3_features), or it is a widening aggregate-first layer that can run on the cached aggregate --
        63 / 313 / 363 -> 1000, 63 -> 206 of the reference's runs: without the cache such a layer copies its fp32 rows per batch and
        makes both images per step."""
        return self.wants_p3_features(f0) or self.wants_agg_image(f0)

    def _layer_kind(self, i: int, L, fin: int, n: int = 0, cached: bool = False):
        """How hidden layer i runs on the one-call plan (gte_step_layer.kind): 0 planes layer in transform-first order, 1 the
        one-pass short-input layer (BBOX features), 2 aggregate-first planes input layer (fin < fout), 3 (``cached``: the batch
        brings the resident image of the input's mean aggregate) the input layer on [x | ahn] from two resident images; None: not on the plan.
        Every shape the reference's runs produce is covered (run_multiple_train.sh:8-113: hidden 1000, or
        int(calculate_hidden) = 96 ... 218, with F0 = 13 ... 831): hidden widths up to 1024, any input width."""
        fout = L.out_feats
        if not (self._planes_on() and isinstance(L.lynorm, nn.LayerNorm) and L.linear.bias is not None
                and (L.activation is None or _is_relu(L.activation))):
            return None
        # (an input layer on images -- kind 0 or 2 -- moves to the cached form when the batch brings the second image)
        up = 3 if (i == 0 and cached and self._cached_layer0(L, fin)) else 0
        if self._planes_layer(i, L, fin, n):
            return up                                            # the tuned range: 128 <= fout <= 256, fout % 16 == 0
        if (i == 0 and not self._transform_first(L, fin) and fout % 16 == 0
                and bool(self.lib.gte_sage_linear_fwd_fuses_ln(2 * fin, fout)) and not ops.use_tiled(n, fin, None)):
            return 1
        if not self.general_planes or fout > 1024:
            return None
        # transform-first while the layer does not widen by more than a quarter (the aggregation then moves fout columns: 1000
        # against 831 costs less than a per-batch fp32 copy of the input rows, and layer 0 reads the RESIDENT image through the row
        # map); aggregate-first for a widening input layer (13 / 63 / 313 / 363 -> 1000: aggregate fin columns)
        return up if (i > 0 or 4 * fout <= 5 * fin) else (3 if up == 3 else 2)

    def _plan_kinds(self, f0: int, n: int, cached: bool = False):
        """Layer kinds of the one-call step (gte_gcnsage_step) or None when the configuration needs the call-by-call path.
        ``cached``: the batch carries ``agg_p3`` (the resident image of the input's mean aggregate behind its row map)."""
        layers = list(self.model.layers)
        if (not self.use_c_step or not self._planes_on() or len(layers) < 2 or len(layers) > 8 or ops._timers is not None):
            return None
        dims = [f0] + [l.out_feats for l in layers]
        last = len(layers) - 1
        Lo = layers[last]
        if (isinstance(Lo.lynorm, nn.LayerNorm) or Lo.activation is not None or Lo.linear.bias is None or Lo.use_pp
                or not 1 <= Lo.out_feats <= 16 or not self.fused_head):
            return None
        if not self.general_planes and not (self._narrow(Lo, dims[last]) and self._fused_head(last, Lo, dims[last])):
            return None
        kinds = []
        for i, L in enumerate(layers[:-1]):
            k = self._layer_kind(i, L, dims[i], n, cached)
            if k is None:
                return None
            # (the planes GEMMs address their output through 32-bit buffer offsets: [n][2 ld] fp32 must stay below 2 GB)
            if k != 1 and (n + 256) * 2 * _c16(L.out_feats) * 4 >= (1 << 31):
                return None
            kinds.append(k)
        return kinds

    def attach_feature_image(self, g) -> bool:
        """Give a graph that is evaluated again and again (the validation graph of train(): the same graph every epoch,
        model_train.py:246,349-353 of the reference) the P3 image of its features, made ONCE: forward_logits then runs the one-call
        plan on the planes kernels instead of the module path.  False when layer 0 does not take an image."""
        if getattr(g, "feat_p3", None) is not None:
            return True
        x = g.ndata.get('feat')
        if x is None or not x.is_cuda or not self.wants_resident_images(x.shape[1]):
            return False
        x = ops._row_major(x.to(torch.float32))
        g.feat_p3 = ops.p3_from_f32(x)
        if self.wants_agg_image(x.shape[1]):
            # ... and the image of the input's mean aggregate (constant too): layer 0 then is ONE launch, [x | ahn] W^T with LayerNorm
            # + ReLU in its epilogue, instead of a GEMM and an aggregation + LayerNorm launch over [n, 2 hidden]
            csr = g.in_csr()
            g.agg_p3 = ops.spmm_csr_p3(csr.indptr, csr.indices, g.in_weights(g.edata.get("feat")), x, x.shape[0], mean=True)
        return True

    @staticmethod
    def _batch_cached(g) -> bool:
        xp, ap = getattr(g, "feat_p3", None), getattr(g, "agg_p3", None)
        return xp is not None and ap is not None and (xp.row_map is None) == (ap.row_map is None)

    def _plan_mode(self, kinds, f0: int):
        """(general, out_gemm) of a plan: ``general`` = it runs on the padded buffer set (_alloc_gen) -- some hidden layer lies
        outside the tuned range or the output layer runs on the planes GEMMs; ``out_gemm`` = the latter (hidden width beyond the
        narrow kernels: > 256 or not a multiple of 8)."""
        layers = list(self.model.layers)
        dims = [f0] + [l.out_feats for l in layers]
        last = len(layers) - 1
        out_gemm = not (self._narrow(layers[last], dims[last]) and
                        (self._fused_head(last, layers[last], dims[last]) or self._narrow_padded(layers[last], dims[last])))
        gen = out_gemm or any(k in (2, 3) or (k == 0 and not self._planes_layer(i, layers[i], dims[i])) for i, k in enumerate(kinds))
        return gen, out_gemm

    def _weight_images_gen(self, dims, kinds, out_gemm: bool):
        """Weight images of a general plan (kept per (kinds, out_gemm)) and their conversion descriptors.  A planes layer's forward
        image holds [W_s rows ; zero rows up to ld ; W_n rows ; zero rows] (ld = fout rounded up to 16: the two halves of t start on
        16-column boundaries), its backward image [fin] x [W_s^T | W_n^T] with the second segment at column block ld / 16; an
        aggregate-first layer's image is [fout] x [W_s | W_n] with the second K segment at block ceil(fin / 16); the output
        layer's images are [32] x [H] (rows 0.. = W_s, 16.. = W_n) and [H] x [32].  Allocated zeroed: the padding is never written."""
        key = ("gen", tuple(dims), tuple(kinds), bool(out_gemm))
        hit = self._wimg.get(key)
        if hit is not None:
            return hit
        dev = self.flat_param.device
        layers = list(self.model.layers)

        def img(rows, cols):                          # (weight images are BLOCK-MAJOR: the B operand of every NT planes GEMM -- ops.P3)
            t = ops.P3.empty(rows, cols, dev, block_major=self.block_major_weights)
            t.data.zero_()
            return t
        imgs, descs = {}, []
        for i, k in enumerate(kinds):
            L = layers[i]
            fin, fout = dims[i], L.out_feats
            W = L.linear.weight
            wp, ldw = W.data_ptr(), W.stride(0)
            if k == 0:
                ld = _c16(fout)
                fwd = img(2 * ld, fin)
                descs.append(_lib.P3Desc(wp, ldw, fout, fin, 0, fwd.at(0, 0), fwd.ldp))
                descs.append(_lib.P3Desc(wp + 4 * fin, ldw, fout, fin, 0, fwd.at(ld, 0), fwd.ldp))
                bwd = None
                if i > 0:
                    bwd = img(fin, 2 * ld)
                    descs.append(_lib.P3Desc(wp, ldw, fin, fout, 1, bwd.at(0, 0), bwd.ldp))
                    descs.append(_lib.P3Desc(wp + 4 * fin, ldw, fin, fout, 1, bwd.at(0, ld // 16), bwd.ldp))
                imgs[i] = (fwd, bwd)
            elif k in (2, 3):
                kp = _c16(fin)
                fwd = img(fout, 2 * kp)
                descs.append(_lib.P3Desc(wp, ldw, fout, fin, 0, fwd.at(0, 0), fwd.ldp))
                descs.append(_lib.P3Desc(wp + 4 * fin, ldw, fout, fin, 0, fwd.at(0, kp // 16), fwd.ldp))
                imgs[i] = (fwd, None)
        if out_gemm:
            Lo = layers[-1]
            H, C = dims[-2], Lo.out_feats
            W = Lo.linear.weight
            wp, ldw = W.data_ptr(), W.stride(0)
            fwd, bwd = img(32, H), img(H, 32)
            descs.append(_lib.P3Desc(wp, ldw, C, H, 0, fwd.at(0, 0), fwd.ldp))
            descs.append(_lib.P3Desc(wp + 4 * H, ldw, C, H, 0, fwd.at(16, 0), fwd.ldp))
            descs.append(_lib.P3Desc(wp, ldw, H, C, 1, bwd.at(0, 0), bwd.ldp))
            descs.append(_lib.P3Desc(wp + 4 * H, ldw, H, C, 1, bwd.at(0, 1), bwd.ldp))
            imgs["out"] = (fwd, bwd)
        if len(descs) > 32:
            raise _lib.GteError("gte_gcnsage_step: more than 32 weight images")
        arr = (_lib.P3Desc * max(len(descs), 1))(*descs)
        hit = self._wimg[key] = (imgs, arr, len(descs))
        return hit

    def _bind_plan_gen(self, g, kinds, with_adam: bool, out_gemm: bool):
        """_bind_plan for a general plan (padded buffer set, _alloc_gen)."""
        lib, P = self.lib, _lib.ptr
        xp = getattr(g, "feat_p3", None)
        if xp is not None:
            x, n, f0 = None, xp.rows, xp.cols
        else:
            x = ops._row_major(g.ndata['feat'])
            _lib.require_device(x, "FusedGcnSageStep")
            n, f0 = x.shape
        b = self._buffers_gen(n, f0, kinds, out_gemm)
        layers = list(self.model.layers)
        dims = [f0] + [l.out_feats for l in layers]
        nh = len(layers) - 1
        ew = g.edata.get("feat")
        csr, rcsr = g.in_csr(), g.out_csr()
        w_in, w_out = g.in_weights(ew), g.out_weights(ew, True)
        plans = b.setdefault("_plans", {})
        cached = plans.get(("gen", with_adam))
        if cached is None:
            imgs, arr, n_desc = self._weight_images_gen(dims, kinds, out_gemm)
            plan = _lib.StepPlan()
            plan.n_hidden = nh
            gs = self._gslice
            for i, L in enumerate(layers[:-1]):
                sl = plan.layer[i]
                fin, fout = dims[i], L.out_feats
                sl.kind, sl.fin, sl.fout, sl.ldf = kinds[i], fin, fout, b["ld"][i]
                sl.W, sl.bias, sl.gamma, sl.beta = P(L.linear.weight), P(L.linear.bias), P(L.lynorm.weight), P(L.lynorm.bias)
                sl.eps, sl.relu = float(L.lynorm.eps), int(L.activation is not None)
                sl.gW, sl.gbias = P(gs[id(L.linear.weight)]), P(gs[id(L.linear.bias)])
                sl.ggamma, sl.gbeta = P(gs[id(L.lynorm.weight)]), P(gs[id(L.lynorm.bias)])
                last_hidden = i == nh - 1
                nxt_img = (not last_hidden and kinds[i + 1] == 0) or (last_hidden and out_gemm)
                # the layer's output: as an image for a planes consumer, as fp32 rows for the narrow output kernels
                if nxt_img:
                    yp = b["hp_out"] if last_hidden else b["hp"][i + 1]
                    sl.yp, sl.ldp_y = P(yp.data), yp.ldp
                sl.y = P(b["y"][i]) if (last_hidden and not out_gemm) or not nxt_img else None
                sl.t, sl.stats, sl.dy = P(b["t"][i]), P(b["stats"][i]), P(b["dy"][i])
                sl.ws_ln, sl.ws_ln_bytes = P(b["ws_ln"][i]), b["ws_ln"][i].numel()
                sl.ws_dw, sl.ws_dw_bytes = P(b["ws_dw"][i]), b["ws_dw"][i].numel()
                if kinds[i] == 1:
                    sl.ahn = P(b["ahn"][i])
                    continue
                wf, wb = imgs[i]
                sl.wimg_fwd, sl.ldp_wfwd = P(wf.data), wf.ldp
                if wb is not None:
                    sl.wimg_bwd, sl.ldp_wbwd = P(wb.data), wb.ldp
                sl.dzp, sl.ldp_o = P(b["dzp"][i].data), b["dzp"][i].ldp
                if kinds[i] == 3:
                    continue                                      # (both operand images are the batch's: bound per call)
                sl.hp, sl.ldp_h = P(b["hp"][i].data), b["hp"][i].ldp
                if kinds[i] == 0:
                    sl.qp = P(b["qp"][i].data)
                else:
                    sl.ahnp, sl.ldp_ahn = P(b["ahnp"][i].data), b["ahnp"][i].ldp
            Lo = layers[-1]
            C = Lo.out_feats
            plan.out_fin, plan.n_classes = dims[-2], C
            plan.W_out, plan.b_out = P(Lo.linear.weight), P(Lo.linear.bias)
            plan.gW_out, plan.gb_out = P(gs[id(Lo.linear.weight)]), P(gs[id(Lo.linear.bias)])
            plan.ld_h_out = b["ld"][-1]
            plan.dh_out = P(b["dy"][-1])
            if out_gemm:
                wf, wb = imgs["out"]
                plan.out_gemm, plan.ld_lg = 1, 32
                plan.hp_out, plan.ldp_hout = P(b["hp_out"].data), b["hp_out"].ldp
                plan.wimg_out_fwd, plan.ldp_wout_fwd = P(wf.data), wf.ldp
                plan.wimg_out_bwd, plan.ldp_wout_bwd = P(wb.data), wb.ldp
                plan.logits, plan.tn = P(b["t_out"]), P(b["t_out"]) + 64
                plan.dl, plan.q_out = P(b["dlq"]), P(b["dlq"]) + 64
                plan.dlqp, plan.ldp_dlq = P(b["dlqp"].data), b["dlqp"].ldp
                plan.ws_out, plan.ws_out_bytes = P(b["ws_out"]), b["ws_out"].numel()
                plan.ws_ce, plan.ws_ce_bytes = P(b["ws_ce"]), b["ws_ce"].numel()
                plan.ws_cs, plan.ws_cs_bytes = P(b["ws_cs"]), b["ws_cs"].numel()
            else:
                plan.h_out = P(b["y"][-1])
                plan.logits, plan.tn, plan.q_out, plan.dl = P(b["logits"]), P(b["tn"]), P(b["q"]), P(b["dl"])
                plan.ce_part, plan.ce_part_bytes = P(b["ce_part"]), b["ce_part"].numel()
                plan.ws_nar, plan.ws_nar_bytes = P(b["ws_nar"]), b["ws_nar"].numel()
            plan.out3 = P(b["out3"])
            plan.wimg_descs, plan.n_wimg_descs = ctypes.addressof(arr), n_desc
            if with_adam:
                plan.param, plan.grad, plan.exp_avg, plan.exp_avg_sq = (P(self.flat_param), P(self.flat_grad), P(self.exp_avg),
                                                                        P(self.exp_avg_sq))
                plan.n_param = self.flat_param.numel()
                plan.hyper, plan.step_counter, plan.ticket = P(self._hyper), P(self._step_dev), P(self._ticket)
            if self._tail_ws is None:
                self._tail_ws = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=self.flat_param.device)
            if self.tail_split:
                plan.tail_ws, plan.tail_ws_bytes = P(self._tail_ws), self._tail_ws.numel()
            cached = plans[("gen", with_adam)] = (plan, arr, ctypes.c_int(0))
        plan, _arr, fused = cached
        # per call: switches and class weights (public attributes), the graph, the features
        plan.class_weights = P(self.class_weights)
        plan.fuse_ln_dx = (int(self.fuse_ln_dx) | (2 if self.fuse_ln_narrow else 0) | (8 if self.fuse_smallk_dx else 0)
                           | (4 if self.fuse_head_gemm else 0) | (16 if self.fuse_ln_fwd else 0))
        L0 = plan.layer[0]
        if kinds[0] == 3:
            ap = getattr(g, "agg_p3", None)
            if xp is None or ap is None:
                raise _lib.GteError("a cached-aggregate input layer needs feat_p3 and agg_p3 (resident images behind a row map, or the graph's own)")
            L0.hp, L0.ldp_h, L0.make_hp, L0.x = P(xp.data), xp.ldp, 0, None
            L0.ahnp, L0.ldp_ahn = P(ap.data), ap.ldp
            L0.h_rows, L0.n_res_rows = (P(xp.row_map), xp.res_rows) if xp.row_map is not None else (None, 0)
        elif kinds[0] == 0:
            if xp is not None:
                L0.hp, L0.ldp_h, L0.make_hp, L0.x = P(xp.data), xp.ldp, 0, None
                L0.h_rows, L0.n_res_rows = (P(xp.row_map), xp.res_rows) if xp.row_map is not None else (None, 0)
            else:
                L0.hp, L0.ldp_h, L0.make_hp = P(b["hp"][0].data), b["hp"][0].ldp, 1
                L0.h_rows, L0.n_res_rows = None, 0
                L0.x, L0.ldx = P(x), ops._ld(x)
        else:
            if x is None:
                # an image-only batch on a layer that reads fp32 rows (copied image rows without the aggregate image, GTE_P3_ROWS=0):
                # the rows back from the image -- exactly the fp32 values; slow, a measurement configuration
                if 'feat' not in g.ndata:
                    g.ndata['feat'] = ops.p3_to_f32(xp)
                x = ops._row_major(g.ndata['feat'])
            L0.x, L0.ldx = P(x), ops._ld(x)
        plan.indptr, plan.indices, plan.w_in = P(csr.indptr), P(csr.indices), P(w_in)
        plan.rindptr, plan.rindices, plan.w_out = P(rcsr.indptr), P(rcsr.indices), P(w_out)
        plan.n_nodes = n
        C = layers[-1].out_feats
        view = {"out3": b["out3"], "_full": b, "_wkey": ("gen", tuple(dims), tuple(kinds), bool(out_gemm)),
                "logits": b["t_out"][:n, :C] if out_gemm else b["logits"][:n]}
        return plan, fused, view, n, (csr, rcsr, w_in, w_out, self.class_weights)

    def _bind_plan(self, g, kinds, with_adam: bool):
        """The gte_step_plan of this layer plan (cached with the buffer set whose addresses it holds) with the per-batch fields --
        graph, features, node count -- set for ``g``.  Returns (plan, fused flag, row views of the buffers, node count, tensors
        the plan points at)."""
        lib, P = self.lib, _lib.ptr
        xp = getattr(g, "feat_p3", None)
        if xp is not None:
            x, n, f0 = None, xp.rows, xp.cols
        else:
            x = ops._row_major(g.ndata['feat'])
            _lib.require_device(x, "FusedGcnSageStep")
            n, f0 = x.shape
        gen, out_gemm = self._plan_mode(kinds, f0)
        if gen:
            return self._bind_plan_gen(g, kinds, with_adam, out_gemm)
        b = self._buffers(n, f0, self._private_key)
        layers = list(self.model.layers)
        dims = [f0] + [l.out_feats for l in layers]
        ew = g.edata.get("feat")
        csr, rcsr = g.in_csr(), g.out_csr()
        w_in, w_out = g.in_weights(ew), g.out_weights(ew, True)
        plans = b["_full"].setdefault("_plans", {})       # cached with the buffer set whose addresses they hold
        key = (tuple(kinds), with_adam)
        cached = plans.get(key)
        if cached is None:
            self._weight_images(dims, launch=False)
            descs = self._wimg_descs
            if len(descs) > 16:
                raise _lib.GteError("gte_gcnsage_step: more than 16 weight images")
            arr = (_lib.P3Desc * max(len(descs), 1))(*descs)
            plan = _lib.StepPlan()
            plan.n_hidden = len(layers) - 1
            for i, L in enumerate(layers[:-1]):
                sl = plan.layer[i]
                fin, fout = dims[i], L.out_feats
                sl.kind, sl.fin, sl.fout = kinds[i], fin, fout
                sl.W, sl.bias, sl.gamma, sl.beta = P(L.linear.weight), P(L.linear.bias), P(L.lynorm.weight), P(L.lynorm.bias)
                sl.eps, sl.relu = float(L.lynorm.eps), int(L.activation is not None)
                gs = self._gslice
                sl.gW, sl.gbias = P(gs[id(L.linear.weight)]), P(gs[id(L.linear.bias)])
                sl.ggamma, sl.gbeta = P(gs[id(L.lynorm.weight)]), P(gs[id(L.lynorm.bias)])
                nxt_planes = i + 1 < len(layers) - 1 and kinds[i + 1] == 0
                if kinds[i] == 0:
                    wf, wb = self._wimg[i]
                    sl.wimg_fwd, sl.ldp_wfwd = P(wf.data), wf.ldp
                    if wb is not None:
                        sl.wimg_bwd, sl.ldp_wbwd = P(wb.data), wb.ldp
                    sl.hp, sl.ldp_h = P(b["hp"][i].data), b["hp"][i].ldp
                    sl.t = P(b["t"][i])
                    sl.dzp, sl.qp, sl.ldp_o = P(b["dzp"][i].data), P(b["qp"][i].data), b["dzp"][i].ldp
                    sl.ws_dw, sl.ws_dw_bytes = P(b["ws_p3"][i]), b["ws_p3"][i].numel()
                    sl.y = None if nxt_planes else P(b["y"][i])
                else:
                    sl.ahn, sl.t = P(b["ahn"][i]), P(b["z"][i])
                    sl.y = None if (nxt_planes and fout % 16 == 0) else P(b["y"][i])
                    sl.ws_dw, sl.ws_dw_bytes = P(b["ws_dw"][i]), b["ws_dw"][i].numel()
                if nxt_planes:
                    sl.yp, sl.ldp_y = P(b["hp"][i + 1].data), b["hp"][i + 1].ldp
                sl.stats, sl.dy = P(b["stats"][i]), P(b["dy"][i])
                sl.ws_ln, sl.ws_ln_bytes = P(b["ws_ln"][i]), b["ws_ln"][i].numel()
            Lo = layers[-1]
            plan.out_fin, plan.n_classes = dims[-2], Lo.out_feats
            plan.W_out, plan.b_out = P(Lo.linear.weight), P(Lo.linear.bias)
            plan.gW_out, plan.gb_out = P(self._gslice[id(Lo.linear.weight)]), P(self._gslice[id(Lo.linear.bias)])
            plan.h_out, plan.ld_h_out = P(b["y"][-2]), dims[-2]
            plan.logits, plan.tn, plan.q_out = P(b["y"][-1]), P(b["tn"]), P(b["q"])
            plan.dl, plan.dh_out = P(b["dy"][-1]), P(b["dy"][-2])
            plan.ce_part, plan.ce_part_bytes = P(b["ce_part"]), b["ce_part"].numel()
            plan.ws_nar, plan.ws_nar_bytes = P(b["ws_nar"]), b["ws_nar"].numel()
            plan.out3 = P(b["out3"])
            plan.wimg_descs, plan.n_wimg_descs = ctypes.addressof(arr), len(descs)
            if with_adam:
                plan.param, plan.grad, plan.exp_avg, plan.exp_avg_sq = (P(self.flat_param), P(self.flat_grad), P(self.exp_avg),
                                                                        P(self.exp_avg_sq))
                plan.n_param = self.flat_param.numel()
                plan.hyper, plan.step_counter, plan.ticket = P(self._hyper), P(self._step_dev), P(self._ticket)
            if self._tail_ws is None:
                self._tail_ws = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=self.flat_param.device)
            if self.tail_split:
                plan.tail_ws, plan.tail_ws_bytes = P(self._tail_ws), self._tail_ws.numel()
            cached = plans[key] = (plan, arr, ctypes.c_int(0))
        plan, _arr, fused = cached
        # per call: the public switches and the class weights (re-read like the call-by-call path does) ...
        plan.class_weights = P(self.class_weights)
        plan.fuse_ln_dx = (int(self.fuse_ln_dx) | (2 if self.fuse_ln_narrow else 0)
                           | (8 if self.fuse_smallk_dx else 0))
        # ... and per batch: the graph, the features, the labels
        b["logits"], b["_wkey"] = b["y"][-1], "tuned"
        L0 = plan.layer[0]
        if kinds[0] == 0:
            if xp is not None:
                L0.hp, L0.ldp_h, L0.make_hp, L0.x = P(xp.data), xp.ldp, 0, None
                L0.h_rows, L0.n_res_rows = (P(xp.row_map), xp.res_rows) if xp.row_map is not None else (None, 0)
            else:
                L0.hp, L0.ldp_h, L0.make_hp = P(b["hp"][0].data), b["hp"][0].ldp, 1
                L0.h_rows, L0.n_res_rows = None, 0
                L0.x, L0.ldx = P(x), ops._ld(x)
        else:
            if x is None:
                raise _lib.GteError("the batch holds its features as a P3 image, but layer 0 reads fp32 rows")
            L0.x, L0.ldx = P(x), ops._ld(x)
        plan.indptr, plan.indices, plan.w_in = P(csr.indptr), P(csr.indices), P(w_in)
        plan.rindptr, plan.rindices, plan.w_out = P(rcsr.indptr), P(rcsr.indices), P(w_out)
        plan.n_nodes = n
        return plan, fused, b, n, (csr, rcsr, w_in, w_out, self.class_weights)

    def _c_step(self, g, labels, grad_scale, kinds, with_adam: bool):
        """forward + loss + backward (+ Adam inside the fold launch) through gte_gcnsage_step: two host calls (the next batch's
        assembly is queued between them) instead of ~19."""
        lib, P = self.lib, _lib.ptr
        st = _lib.current_stream()
        plan, fused, b, n, keep = self._bind_plan(g, kinds, with_adam)
        _arr = plan.wimg_descs                                       # (the descriptor array lives in the plan cache)
        lab = labels if labels.dtype in (torch.float32, torch.int64) else labels.to(torch.int64)
        plan.labels, plan.labels_f32 = P(lab), int(lab.dtype == torch.float32)
        plan.grad_scale = float(grad_scale)
        if self.fwd_events:
            for e in self.fwd_events:
                if not e.cuda_event:                  # (created lazily by the first record)
                    e.record()
            self._fwd_ev_arr = (ctypes.c_void_p * len(self.fwd_events))(*[e.cuda_event for e in self.fwd_events])
            plan.fwd_events = ctypes.addressof(self._fwd_ev_arr)
        else:
            plan.fwd_events = None
        capturing = torch.cuda.is_current_stream_capturing()
        sig = self._param_sig()
        sig = (sig, b["_wkey"])                   # (which image set: a plan of another layout keeps its own images)
        plan.wimg_fresh = int(self.wimg_in_fold and not capturing and self._wimg_sig == sig)
        plan.wimg_in_fold = int(self.wimg_in_fold and with_adam and not capturing)
        self._wimg_sig = None
        addr = ctypes.addressof(plan)
        if self.before_last_gemm is not None:
            _lib.check(lib.gte_gcnsage_step(addr, 1, ctypes.byref(fused), st), "gte_gcnsage_step")
            # phase 1 returned with this thread's fold deferral OPEN and the tail workspace registered: whatever the callback (the
            # next batch's assembly) or phase 2 raises, both are closed again -- a step that died here must not poison the next
            # one ("a deferral is already open") or let later standalone calls queue folds nobody flushes
            try:
                self.before_last_gemm()
                _lib.check(lib.gte_gcnsage_step(addr, 2, ctypes.byref(fused), st), "gte_gcnsage_step")
            except BaseException:
                lib.gte_fold_defer_flush()                      # (an error if phase 2 already closed it: ignored)
                lib.gte_gemm_set_tail_workspace(None, 0)
                self._wimg_sig = None
                raise
        else:
            _lib.check(lib.gte_gcnsage_step(addr, 0, ctypes.byref(fused), st), "gte_gcnsage_step")
        self._adam_fused = bool(fused.value & 1)
        # the images now hold: the updated parameters (written by the fold launch), or -- no optimiser step in this call -- the
        # unchanged ones the forward converted; otherwise Adam follows as its own launch and they are stale
        if not capturing and ((fused.value & 2) or not with_adam):
            self._wimg_sig = sig
        self._keep = (lab,) + tuple(keep)                              # alive until the next step
        # (the data-parallel step's own Adam launch writes the same images: gte_adam_step_dev_images)
        # (kept only while an UNFUSED optimiser launch may still follow this very step -- a step whose fold launch already ran Adam
        # must not leave descriptors behind for a later launch of another plan; the tuple keeps the descriptor array AND the image
        # tensors it points into alive)
        self._last_wimg = ((plan.wimg_descs, int(plan.n_wimg_descs), b["_wkey"], _arr, dict(self._wimg))
                           if not capturing and not (fused.value & 1) else None)
        return b["out3"]

    FORWARD_IMAGE_MAX_ELEMS = 1 << 21     # forward_logits: largest fp32 feature matrix that is converted to an image per call

    def forward_logits(self, g) -> torch.Tensor:
        """``model(g)`` without autograd (model_predict.py:141-147, the validation forward of model_train.py:349-353): logits
        [n, n_classes] through ONE host call (gte_gcnsage_forward) on the step's own buffers -- a view that the next step or
        forward on this engine overwrites.  Configurations the one-call plan does not cover -- and large graphs that bring fp32
        features to a planes input layer (a validation graph: the plan would write their image first) -- run the module path."""
        self._last_wimg = None
        xp = getattr(g, "feat_p3", None)
        n, f0 = (xp.rows, xp.cols) if xp is not None else g.ndata['feat'].shape
        kinds = self._plan_kinds(f0, n, self._batch_cached(g)) if n > 0 else None
        if kinds is not None and xp is None and kinds[0] == 0 and n * f0 > self.FORWARD_IMAGE_MAX_ELEMS:
            # fp32 features under a planes input layer: the one-call plan would first write their P3 image (65 us at 21.5 k x 831)
            # -- more than the call saves on a graph of this size; the module path multiplies the fp32 rows directly
            # (profiles/debug/val_forward_time.py: 0.241 against 0.275 ms at 21.5 k nodes, 1.24 against 1.43 ms at 124 k)
            kinds = None
        if kinds is None:
            if xp is not None and 'feat' not in g.ndata:
                # a resident batch in image mode carries its features as a P3 image only: the module path reads fp32 rows (the
                # image holds exactly the fp32 values)
                g.ndata['feat'] = ops.p3_to_f32(xp)
            with torch.no_grad():
                return self.model(g)
        plan, _fused, b, n, keep = self._bind_plan(g, kinds, with_adam=False)
        capturing = torch.cuda.is_current_stream_capturing()
        sig = (self._param_sig(), b["_wkey"])
        plan.wimg_fresh = int(self.wimg_in_fold and not capturing and self._wimg_sig == sig)
        _lib.check(self.lib.gte_gcnsage_forward(ctypes.addressof(plan), _lib.current_stream()), "gte_gcnsage_forward")
        self._wimg_sig = None if capturing else sig
        self._keep = keep
        return b["logits"]

    # -- the schedule ----------------------------------------------------------------------------------
    def forward_backward(self, g, labels: torch.Tensor, grad_scale: float = 1.0, upto_layer: int = 0) -> torch.Tensor:
        """Forward, loss and the backward of layers n_layers-1 .. upto_layer (gradients of those layers final on return:
        their folds are flushed).  upto_layer > 0 leaves the rest to :meth:`backward_rest` -- the data-parallel step
        all-reduces the upper layers' gradient slice while the (longest) backward of layer 0 runs."""
        if upto_layer == 0:
            xp = getattr(g, "feat_p3", None)
            n, f0 = (xp.rows, xp.cols) if xp is not None else g.ndata['feat'].shape
            kinds = self._plan_kinds(f0, n, self._batch_cached(g))
            if kinds is not None:
                return self._c_step(g, labels, grad_scale, kinds, with_adam=bool(self._fuse_adam_req))
        return self._run(g, labels, grad_scale, len(self.model.layers) - 1, upto_layer, forward=True)

    def backward_rest(self, g, from_layer: int) -> None:
        """Backward of layers from_layer-1 .. 0 after ``forward_backward(..., upto_layer=from_layer)`` on the same batch."""
        self._run(g, None, 1.0, from_layer - 1, 0, forward=False)

    def _run(self, g, labels, grad_scale, hi, lo, forward):
        lib, P, check = self.lib, _lib.ptr, _lib.check
        self._wimg_sig = None                         # (this schedule converts the weight images in front of every forward)
        self._last_wimg = None                        # (... and no later optimiser launch may rewrite another plan's images)
        st = _lib.current_stream()
        timed = ops._timed
        xp = getattr(g, "feat_p3", None)              # resident batches in image mode bring the features as a P3 image only
        if xp is not None:
            x, n, f0 = None, xp.rows, xp.cols
            if not self._planes_layer(0, self.model.layers[0], f0, n):
                # an image batch on a layer that reads fp32 rows (this schedule runs layer 0 on planes for fewer shapes than the
                # one-call plans, and does not know the cached-aggregate form; or the GEMM mode changed after the resident pages
                # were converted): the rows back from the image -- exactly the fp32 values
                if 'feat' not in g.ndata:
                    g.ndata['feat'] = ops.p3_to_f32(xp)
                x, xp = ops._row_major(g.ndata['feat']), None
        else:
            x = ops._row_major(g.ndata['feat'])
            _lib.require_device(x, "FusedGcnSageStep")
            n, f0 = x.shape
        b = self._buffers(n, f0, self._private_key)
        b["xp"] = xp
        # (kept with the buffer set, not with this call's row views: backward_rest() of the data-parallel overlap reads what the
        # forward of forward_backward() left)
        b["hp_used"] = b["_full"].setdefault("_hp_used", [None] * len(self.model.layers))
        layers = list(self.model.layers)
        ew = g.edata.get("feat")
        csr, rcsr = g.in_csr(), g.out_csr()
        w_in, w_out = g.in_weights(ew), g.out_weights(ew, True)
        if forward and any(b["pl"]):
            self._weight_images([f0] + [l.out_feats for l in layers])
        big = n * max(f0, max(l.out_feats for l in self.model.layers)) * 4 >= min(ops.TILED_FULL_MIN_BYTES, ops.TILED_MIN_BYTES)
        t_in, t_out = (g.in_tiles(), g.out_tiles()) if big else (None, None)

        PP = lambda a: a if isinstance(a, int) else P(a)           # tensor or raw device address (a column offset into one)

        def aggregate(csr_, w_, tiles_, src, ldsrc, dst, lddst, f, reduce, accumulate):
            nbytes = 2.0 * n * f * 4 + 8.0 * csr_.indices.numel() + 4.0 * (n + 1)
            if tiles_ is not None and ops.use_tiled(n, f, csr_.indices.numel()):
                with timed("spmm_tiled", nbytes):
                    check(lib.gte_spmm_csr_tiled(P(csr_.indptr), P(csr_.indices), P(tiles_.local_index), P(w_),
                                                 P(tiles_.tile_ptr), P(tiles_.tile_src), PP(src), ldsrc, PP(dst), lddst,
                                                 n, f, reduce, int(accumulate), st), "gte_spmm_csr_tiled")
            else:
                fn = lib.gte_spmm_csr_accumulate if accumulate else lib.gte_spmm_csr
                with timed("spmm_csr", nbytes):
                    check(fn(P(csr_.indptr), P(csr_.indices), P(w_), PP(src), ldsrc, PP(dst), lddst, n, f, _lib.GTE_F32,
                             reduce, st), "gte_spmm_csr")
        # scratch for the GEMM tail split (see gte_gemm_set_tail_workspace): registered for this launch sequence only
        if self._tail_ws is None:
            self._tail_ws = torch.empty(int(lib.gte_gemm_tail_workspace_bytes()), dtype=torch.uint8, device=self.flat_param.device)
        check(lib.gte_gemm_set_tail_workspace(P(self._tail_ws) if self.tail_split else None,
                                              self._tail_ws.numel() if self.tail_split else 0), "gte_gemm_set_tail_workspace")
        try:
            if forward:
                self._forward_loss(g, labels, grad_scale, x, n, f0, b, layers, csr, w_in, t_in, aggregate, st)
                self._ln_p3_done = None
                self._smallk_done = False
            # ---------------- backward of layers hi .. lo ----------------
            check(lib.gte_fold_defer_begin(st), "gte_fold_defer_begin")
            try:
                self._backward(g, b, layers, x, n, aggregate, rcsr, w_out, t_out, st, hi, lo)
            finally:
                if self._fuse_adam_req and lo == 0:
                    # the folds produce every gradient element: the optimiser step rides in the same launch (falls back to a
                    # plain flush, fused = 0, when some gradient was written directly)
                    fused = ctypes.c_int(0)
                    check(lib.gte_fold_defer_flush_adam(P(self.flat_param), P(self.flat_grad), P(self.exp_avg), P(self.exp_avg_sq),
                                                        self.flat_param.numel(), P(self._hyper), P(self._step_dev),
                                                        P(self._ticket), ctypes.byref(fused)), "gte_fold_defer_flush_adam")
                    self._adam_fused = bool(fused.value)
                else:
                    check(lib.gte_fold_defer_flush(), "gte_fold_defer_flush")
            return b["out3"]
        finally:
            lib.gte_gemm_set_tail_workspace(None, 0)

    def _forward_loss(self, g, labels, grad_scale, x, n, f0, b, layers, csr, w_in, t_in, aggregate, st):
        lib, P, check = self.lib, _lib.ptr, _lib.check
        timed, ld = ops._timed, ops._ld
        ws, wsn = P(b["ws"]), b["ws"].numel()
        # ---------------- forward ----------------
        h = x
        b["hp_used"][:] = [None] * len(layers)
        fused_head = False
        pending_ln = None            # (layer, z, y, stats) of a LayerNorm left to the output layer's forward kernel
        hp_in = None                 # P3 image of the current layer's input (set by the producer of h)
        for i, L in enumerate(layers):
            fin, fout = (f0 if i == 0 else layers[i - 1].out_feats), L.out_feats
            W, bias = L.linear.weight, L.linear.bias
            ln = isinstance(L.lynorm, nn.LayerNorm)
            relu = L.activation is not None
            ahn, y = b["ahn"][i], b["y"][i]
            if b["pl"][i]:
                # ---- planes layer: t = h [W_s ; W_n]^T + [b | 0] (planes GEMM), then z = t_self + mean-aggregate(t_neigh),
                # LayerNorm, ReLU in ONE pass that writes y as the next planes layer's input image (and / or fp32)
                if hp_in is None:
                    hp_in = b["xp"] if (i == 0 and b["xp"] is not None) else b["hp"][i]
                    if not (i == 0 and b["xp"] is not None):
                        check(lib.gte_p3_from_f32(P(h), ld(h), n, fin, 0, P(hp_in.data), hp_in.ldp, st), "gte_p3_from_f32")
                b["hp_used"][i] = hp_in
                wf = self._wimg[i][0]
                t = b["t"][i]
                with timed("gemm_nt", 4.0 * n * fin * fout) as tm:
                    for _ in tm.repeat():
                        if hp_in.row_map is not None:      # the RESIDENT image through the batch's row map
                            check(lib.gte_gemm_p3_nt_rows(P(hp_in.data), hp_in.ldp, fin, P(hp_in.row_map), hp_in.res_rows, P(wf.data),
                                                          wf.ldp, P(bias), fout, P(t), 2 * fout, n, 2 * fout, 0, 0, st),
                                  "gte_gemm_p3_nt_rows")
                        else:
                            check(lib.gte_gemm_p3_nt(P(hp_in.data), hp_in.ldp, fin, None, 0, 0, P(wf.data), wf.ldp, P(bias), fout, P(t),
                                                     2 * fout, n, 2 * fout, 0, 0, st), "gte_gemm_p3_nt")
                nxt_planes = i + 1 < len(layers) and b["pl"][i + 1]
                yp = b["hp"][i + 1] if nxt_planes else None
                with timed("spmm_csr", 3.0 * n * fout * 4 + 8.0 * csr.indices.numel() + 4.0 * (n + 1)):
                    check(lib.gte_spmm_csr_accumulate_ln_p3(P(csr.indptr), P(csr.indices), P(w_in), P(t) + 4 * fout, 2 * fout, P(t),
                                                            2 * fout, n, fout, _lib.REDUCE_MEAN, P(L.lynorm.weight),
                                                            P(L.lynorm.bias), float(L.lynorm.eps), int(relu),
                                                            None if nxt_planes else P(y), fout,
                                                            P(yp.data) if yp is not None else None, yp.ldp if yp is not None else 0,
                                                            P(b["stats"][i]), st), "gte_spmm_csr_accumulate_ln_p3")
                h, hp_in = y, yp
                continue
            hp_in = None
            if self._narrow(L, fin):
                # class-count-wide layer: logits = h W_s^T + b + mean-aggregate(h W_n^T)  (aggregation on C columns)
                with timed("narrow_fwd", 2.0 * n * fin * 4):
                    if pending_ln is not None:
                        # the layer below left its pre-LayerNorm z: normalise, write y / stats and multiply in one pass
                        Lb, zb, yb, sb = pending_ln
                        check(lib.gte_sage_narrow_fwd_ln(P(zb), ld(zb), fin, P(Lb.lynorm.weight), P(Lb.lynorm.bias),
                                                         float(Lb.lynorm.eps), int(Lb.activation is not None), P(yb), fin, P(sb),
                                                         P(W), 2 * fin, P(bias), fout, P(y), fout, P(b["tn"]), fout, n, st),
                              "gte_sage_narrow_fwd_ln")
                        pending_ln = None
                    else:
                        check(lib.gte_sage_narrow_fwd(P(h), ld(h), fin, P(W), 2 * fin, P(bias), fout, P(y), fout, P(b["tn"]),
                                                      fout, n, st), "gte_sage_narrow_fwd")
                fused_head = self._fused_head(i, L, fin)
                if not fused_head:
                    aggregate(csr, w_in, None, b["tn"], fout, y, fout, fout, _lib.REDUCE_MEAN, True)
                h = y
                continue
            if self._transform_first(L, fin):
                t = b["t"][i]
                with timed("gemm_nt", 4.0 * n * fin * fout) as tm:
                    for _ in tm.repeat():
                        check(lib.gte_sage_transform_fwd(P(h), ld(h), fin, P(W), 2 * fin, P(bias), fout, P(t), 2 * fout, n,
                                                         st), "gte_sage_transform_fwd")
                if (fout % 4 == 0 and lib.gte_spmm_csr_accumulate_ln_supported(fout)
                        and not (t_in is not None and ops.use_tiled(n, fout, csr.indices.numel(), fused_ln=True))):
                    # z = t_self + mean-aggregate(t_neigh) and y = relu(LayerNorm(z)) in one pass over the rows
                    with timed("spmm_csr", 3.0 * n * fout * 4 + 8.0 * csr.indices.numel() + 4.0 * (n + 1)):
                        check(lib.gte_spmm_csr_accumulate_ln(P(csr.indptr), P(csr.indices), P(w_in), P(t) + 4 * fout, 2 * fout,
                                                             P(t), 2 * fout, n, fout, _lib.REDUCE_MEAN, P(L.lynorm.weight),
                                                             P(L.lynorm.bias), float(L.lynorm.eps), int(relu), P(y), fout,
                                                             P(b["stats"][i]), st), "gte_spmm_csr_accumulate_ln")
                else:
                    aggregate(csr, w_in, t_in, P(t) + 4 * fout, 2 * fout, t, 2 * fout, fout, _lib.REDUCE_MEAN, True)
                    check(lib.gte_ln_relu_fwd(P(t), 2 * fout, P(L.lynorm.weight), P(L.lynorm.bias), float(L.lynorm.eps),
                                              int(relu), P(y), fout, P(b["stats"][i]), n, fout, st), "gte_ln_relu_fwd")
                h = y
                continue
            aggregate(csr, w_in, t_in, h, ld(h), ahn, fin, fin, _lib.REDUCE_MEAN, False)
            if ln and lib.gte_sage_linear_fwd_fuses_ln(2 * fin, fout):
                # short K (BBOX features, 13 + 13 inputs): linear + LayerNorm + ReLU in one pass over the rows; when the next
                # layer is a planes layer its input image is written by the same pass (and y itself is not needed)
                zs = None if self._smallk_bwd(i, L, fin) else P(b["z"][i])   # (the one-pass backward recomputes z)
                if i + 1 < len(layers) and b["pl"][i + 1] and fout % 16 == 0:
                    yp = b["hp"][i + 1]
                    check(lib.gte_sage_linear_fwd_p3(P(h), ld(h), fin, P(ahn), fin, fin, P(W), 2 * fin, P(bias), P(L.lynorm.weight),
                                                     P(L.lynorm.bias), float(L.lynorm.eps), int(relu), zs, fout,
                                                     P(b["stats"][i]), None, fout, P(yp.data), yp.ldp, n, fout, st),
                          "gte_sage_linear_fwd_p3")
                    h, hp_in = y, yp
                    continue
                check(lib.gte_sage_linear_fwd(P(h), ld(h), fin, P(ahn), fin, fin, P(W), 2 * fin, P(bias), P(L.lynorm.weight),
                                              P(L.lynorm.bias), float(L.lynorm.eps), int(relu), zs, fout,
                                              P(b["stats"][i]), P(y), fout, n, fout, st), "gte_sage_linear_fwd")
                h = y
                continue
            lin_out = b["z"][i] if ln else y
            with timed("gemm_nt", 4.0 * n * fin * fout) as tm:
                for _ in tm.repeat():
                    check(lib.gte_sage_linear_fwd(P(h), ld(h), fin, P(ahn), fin, fin, P(W), 2 * fin, P(bias), None, None,
                                                  1e-5, int(relu and not ln), None, 0, None, P(lin_out), fout, n, fout, st),
                          "gte_sage_linear_fwd")
            if ln:
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                if (self.fuse_ln_fwd and nxt is not None and i + 1 == len(layers) - 1 and self._narrow(nxt, fout)
                        and lib.gte_sage_narrow_fwd_ln_supported(fout, nxt.out_feats)):
                    pending_ln = (L, lin_out, y, b["stats"][i])
                else:
                    check(lib.gte_ln_relu_fwd(P(lin_out), fout, P(L.lynorm.weight), P(L.lynorm.bias), float(L.lynorm.eps),
                                              int(relu), P(y), fout, P(b["stats"][i]), n, fout, st), "gte_ln_relu_fwd")
            h = y
        logits = h

        # ---------------- loss ----------------
        lab = labels if labels.dtype in (torch.float32, torch.int64) else labels.to(torch.int64)
        dl = b["dy"][-1]
        self._head_scale = None
        if fused_head:
            # one launch: logits += mean-aggregate(t_neigh), CE terms, UNNORMALISED gradient; 1 / sum(w) is applied (and
            # the loss published) by the output layer's backward kernel -- see gte_head_agg_ce in include/gte.h
            with timed("spmm_csr", 2.0 * n * logits.shape[1] * 4 + 8.0 * csr.indices.numel() + 4.0 * (n + 1)):
                check(lib.gte_head_agg_ce(P(csr.indptr), P(csr.indices), P(w_in), P(b["tn"]), logits.shape[1], P(logits),
                                          logits.shape[1], P(lab), int(lab.dtype == torch.float32), P(self.class_weights), n,
                                          logits.shape[1], _lib.REDUCE_MEAN, P(dl), dl.shape[1], P(b["ce_part"]),
                                          b["ce_part"].numel(), st), "gte_head_agg_ce")
            self._head_scale = float(grad_scale)
        else:
            check(lib.gte_weighted_ce(P(logits), logits.shape[1], P(lab), int(lab.dtype == torch.float32),
                                      P(self.class_weights), n, logits.shape[1], float(grad_scale), P(dl), dl.shape[1],
                                      P(b["out3"]), ws, wsn, st), "gte_weighted_ce")

    def _backward(self, g, b, layers, x, n, aggregate, rcsr, w_out, t_out, st, hi, lo) -> None:
        lib, P, check = self.lib, _lib.ptr, _lib.check
        timed, ld = ops._timed, ops._ld
        ws, wsn = P(b["ws"]), b["ws"].numel()
        for i in range(hi, lo - 1, -1):
            L = layers[i]
            hin = x if i == 0 else b["y"][i - 1]
            fin, fout = (L.linear.weight.shape[1] // 2), L.out_feats
            W = L.linear.weight
            ln = isinstance(L.lynorm, nn.LayerNorm)
            relu = L.activation is not None
            dy = b["dy"][i]
            gW = self._gslice[id(W)]
            gb = self._gslice[id(L.linear.bias)] if L.linear.bias is not None else None
            gg = self._gslice[id(L.lynorm.weight)] if ln else None
            gbe = self._gslice[id(L.lynorm.bias)] if ln else None
            if self._narrow(L, fin):
                # q = A_w^T (norm * dlogits) on C columns; dW = [dl^T h | q^T h], dh = dl W_s + q W_n, dbias = colsum(dl)
                aggregate(rcsr, w_out, None, dy, fout, b["q"], fout, fout, _lib.REDUCE_SUM, False)
                dh = b["dy"][i - 1] if i > 0 else None
                with timed("narrow_bwd", 3.0 * n * fin * 4):
                    if self._ln_rows_below(i, layers, fin, b):
                        # the LayerNorm(+ReLU) backward of the planes layer below on the dh tile of every row block (row form):
                        # d(loss)/d(y) of that layer is never stored, its dz comes out as fp32 + image
                        Lb, gsl, dzb, wsl = layers[i - 1], self._gslice, b["dzp"][i - 1], b["ws_ln"][i - 1]
                        hs = self._head_scale
                        check(lib.gte_sage_narrow_bwd_ln_p3(
                            P(dy), fout, P(b["q"]), fout, P(hin), ld(hin), fin, P(W), 2 * fin, fout, P(dh), fin,
                            P(dzb.data), dzb.ldp,
                            P(gW), 2 * fin, P(gb), n, P(b["ws_nar"]), b["ws_nar"].numel(), P(b["ce_part"]) if hs is not None else None,
                            hs if hs is not None else 1.0, P(b["out3"]) if hs is not None else None, P(b["t"][i - 1]), 2 * fin,
                            P(b["stats"][i - 1]), P(Lb.lynorm.weight), P(Lb.lynorm.bias), int(Lb.activation is not None),
                            P(gsl[id(Lb.lynorm.weight)]), P(gsl[id(Lb.lynorm.bias)]), P(gsl[id(Lb.linear.bias)]), P(wsl), wsl.numel(),
                            st), "gte_sage_narrow_bwd_ln_p3")
                        self._ln_p3_done = i - 1
                    elif self._head_scale is not None and i == len(layers) - 1:
                        check(lib.gte_sage_narrow_bwd_ce(P(dy), fout, P(b["q"]), fout, P(hin), ld(hin), fin, P(W), 2 * fin, fout,
                                                         P(dh), fin, P(gW), 2 * fin, P(gb), n, P(b["ws_nar"]),
                                                         b["ws_nar"].numel(), P(b["ce_part"]), self._head_scale, P(b["out3"]),
                                                         st), "gte_sage_narrow_bwd_ce")
                    else:
                        check(lib.gte_sage_narrow_bwd(P(dy), fout, P(b["q"]), fout, P(hin), ld(hin), fin, P(W), 2 * fin, fout,
                                                      P(dh), fin, P(gW), 2 * fin, P(gb), n, P(b["ws_nar"]),
                                                      b["ws_nar"].numel(), st), "gte_sage_narrow_bwd")
                continue
            if b["pl"][i]:
                # ---- planes layer: dz (fp32 for the transpose aggregation + image), q = A_w^T (norm dz) as an image,
                # dW = [dz^T h | q^T h] and dh = dz W_s + q W_n on the planes GEMMs
                t, dzp, qp, hp = b["t"][i], b["dzp"][i], b["qp"][i], b["hp_used"][i]
                if self._ln_p3_done != i:         # (else: the launch above ran this layer's LayerNorm backward as its epilogue)
                    check(lib.gte_ln_relu_bwd_p3(P(dy), fout, P(t), 2 * fout, P(b["stats"][i]), P(L.lynorm.weight),
                                                 P(L.lynorm.bias), int(relu), P(dy), fout, P(dzp.data), dzp.ldp, P(gg), P(gbe), P(gb),
                                                 n, fout, P(b["ws_ln"][i]), b["ws_ln"][i].numel(), st), "gte_ln_relu_bwd_p3")
                with timed("spmm_csr", 2.0 * n * fout * 4 + 8.0 * rcsr.indices.numel() + 4.0 * (n + 1)):
                    check(lib.gte_spmm_csr_p3(P(rcsr.indptr), P(rcsr.indices), P(w_out), P(dy), fout, P(qp.data), qp.ldp, n, fout,
                                              _lib.REDUCE_SUM, st), "gte_spmm_csr_p3")
                if i == 0 and self.before_last_gemm is not None:
                    self.before_last_gemm()
                wsp = b["ws_p3"][i]

                def dw_planes(stream):
                    if hp.row_map is not None:
                        check(lib.gte_gemm_p3_tn_rows(P(dzp.data), dzp.ldp, P(qp.data), qp.ldp, P(hp.data), hp.ldp, P(hp.row_map),
                                                      hp.res_rows, fin, P(gW), 2 * fin, fout, 2 * fin, n, P(wsp), wsp.numel(), stream),
                              "gte_gemm_p3_tn_rows")
                    else:
                        check(lib.gte_gemm_p3_tn(P(dzp.data), dzp.ldp, P(qp.data), qp.ldp, P(hp.data), hp.ldp, None, 0, fin, P(gW),
                                                 2 * fin, fout, 2 * fin, n, P(wsp), wsp.numel(), stream), "gte_gemm_p3_tn")
                with timed("gemm_tn", 4.0 * n * fin * fout) as tm:
                    for _ in tm.repeat():
                        dw_planes(st)
                if i > 0:
                    wb = self._wimg[i][1]
                    Lb = layers[i - 1]
                    fin_b = Lb.linear.weight.shape[1] // 2
                    if (self.fuse_smallk_dx and i == 1 and self._smallk_bwd(0, Lb, fin_b)
                            and lib.gte_gemm_p3_nt_smallk_bwd_supported(2 * fin_b, fin)):
                        # dX with the WHOLE backward of the short-input layer below as its epilogue: nothing of layer 0 is left
                        gsl, wsd = self._gslice, b["ws_dw"][0]
                        check(lib.gte_gemm_p3_nt_smallk_bwd(P(dzp.data), dzp.ldp, fout, P(qp.data), qp.ldp, fout, P(wb.data), wb.ldp,
                                                            P(x), ld(x), fin_b, P(b["ahn"][0]), fin_b, fin_b, P(Lb.linear.weight),
                                                            2 * fin_b, P(Lb.linear.bias), P(Lb.lynorm.weight), P(Lb.lynorm.bias),
                                                            P(b["stats"][0]), int(Lb.activation is not None), P(gsl[id(Lb.linear.weight)]),
                                                            2 * fin_b, P(gsl[id(Lb.linear.bias)]), P(gsl[id(Lb.lynorm.weight)]),
                                                            P(gsl[id(Lb.lynorm.bias)]), n, fin, P(wsd), wsd.numel(), st),
                              "gte_gemm_p3_nt_smallk_bwd")
                        self._smallk_done = True
                    elif self.fuse_ln_dx and b["pl"][i - 1] and lib.gte_gemm_p3_nt_ln_bwd_supported(fin):
                        # dX with the LayerNorm(+ReLU) backward of the layer below as its epilogue: d(loss)/d(y) of that layer
                        # is never stored, its dz comes out as fp32 + image
                        gsl, dzb, wsl = self._gslice, b["dzp"][i - 1], b["ws_ln"][i - 1]
                        check(lib.gte_gemm_p3_nt_ln_bwd(P(dzp.data), dzp.ldp, fout, P(qp.data), qp.ldp, fout, P(wb.data), wb.ldp,
                                                        P(b["t"][i - 1]), 2 * fin, P(b["stats"][i - 1]), P(Lb.lynorm.weight),
                                                        P(Lb.lynorm.bias), int(Lb.activation is not None), P(b["dy"][i - 1]), fin,
                                                        P(dzb.data), dzb.ldp, P(gsl[id(Lb.lynorm.weight)]), P(gsl[id(Lb.lynorm.bias)]),
                                                        P(gsl[id(Lb.linear.bias)]), n, fin, P(wsl), wsl.numel(), st),
                              "gte_gemm_p3_nt_ln_bwd")
                        self._ln_p3_done = i - 1
                    else:
                        with timed("gemm_nn", 4.0 * n * fin * fout) as tm:
                            for _ in tm.repeat():
                                check(lib.gte_gemm_p3_nt(P(dzp.data), dzp.ldp, fout, P(qp.data), qp.ldp, fout, P(wb.data), wb.ldp, None,
                                                         0, P(b["dy"][i - 1]), fin, n, fin, 0, 0, st), "gte_gemm_p3_nt dX")
                continue
            if self._smallk_bwd(i, L, fin) and self._smallk_done:
                if self.before_last_gemm is not None:                  # (its backward ran as the epilogue of the layer above's dX)
                    self.before_last_gemm()
                continue
            if self._smallk_bwd(i, L, fin):
                # short-input layer 0: LayerNorm(+ReLU) backward and dW in ONE pass over dy (z recomputed, dz never stored)
                if self.before_last_gemm is not None:
                    self.before_last_gemm()
                wdw = b["ws_dw"][i]
                check(lib.gte_sage_smallk_bwd(P(dy), fout, P(hin), ld(hin), fin, P(b["ahn"][i]), fin, fin, P(W), 2 * fin,
                                              P(L.linear.bias), P(L.lynorm.weight), P(L.lynorm.bias), P(b["stats"][i]), int(relu),
                                              P(gW), 2 * fin, P(gb), P(gg), P(gbe), n, fout, P(wdw), wdw.numel(), st),
                      "gte_sage_smallk_bwd")
                continue
            tfirst, qform = self._transform_first(L, fin), self._qform(i, L, fin)
            zsrc = b["t"][i] if tfirst else (b["z"][i] if ln else b["y"][i])
            # dz in place of dy; column sums straight into the flat gradient
            check(lib.gte_ln_relu_bwd(P(dy), fout, P(zsrc), 2 * fout if tfirst else fout, P(b["stats"][i]) if ln else None,
                                      P(L.lynorm.weight) if ln else None, P(L.lynorm.bias) if ln else None, int(relu),
                                      P(dy), fout, P(gg), P(gbe), P(gb), n, fout, P(b["ws_ln"][i]), b["ws_ln"][i].numel(),
                                      st), "gte_ln_relu_bwd")
            dz, ahn = dy, b["ahn"][i]
            if qform:
                # q = A_w^T (norm * dz) into the dead right half of t (transform-first) or the dead ahn buffer
                qp, ldq = (P(b["t"][i]) + 4 * fout, 2 * fout) if tfirst else (P(ahn), fin)
                aggregate(rcsr, w_out, t_out, dz, fout, qp, ldq, fout, _lib.REDUCE_SUM, False)
            # dW is MFMA-bound and nothing downstream needs it before Adam; the rest of the backward chain (dX, the
            # transpose aggregation, the next LayerNorm backward) is mostly HBM-bound: run dW on the side stream so
            # the two kinds of work share the chip.  dz (= dy_i, final for this step) and ahn/h are read-only here.
            wdw = b["ws_dw"][i]
            if i == 0 and self.before_last_gemm is not None:
                self.before_last_gemm()

            def dw_launch(stream):
                if qform:
                    check(lib.gte_sage_qform_dw(P(dz), fout, qp, ldq, P(hin), ld(hin), fin, P(gW), 2 * fin, fout, n, P(wdw),
                                                wdw.numel(), stream), "gte_sage_qform_dw")
                else:
                    check(lib.gte_sage_linear_dw(P(dz), fout, P(hin), ld(hin), fin, P(ahn), fin, fin, P(gW), 2 * fin, fout,
                                                 n, P(wdw), wdw.numel(), stream), "gte_sage_linear_dw")
            with timed("gemm_tn", 4.0 * n * fin * fout) as tm:
                for _ in tm.repeat():
                    dw_launch(st)
            if i > 0 and qform:
                with timed("gemm_nn", 4.0 * n * fin * fout) as tm:
                    for _ in tm.repeat():
                        check(lib.gte_sage_qform_dx(P(dz), fout, qp, ldq, P(W), 2 * fin, fin, fout, P(b["dy"][i - 1]), fin, n,
                                                    st), "gte_sage_qform_dx")
            elif i > 0:
                dh, dahn = b["dy"][i - 1], b["dahn"]
                with timed("gemm_nn", 4.0 * n * fin * fout):
                    check(lib.gte_gemm_f32(0, 0, n, fin, fout, P(dz), fout, P(W), 2 * fin, P(dh), fin, 0, ws, wsn, st),
                          "gte_gemm_f32 dh_self")
                    check(lib.gte_gemm_f32(0, 0, n, fin, fout, P(dz), fout, P(W) + 4 * fin, 2 * fin, P(dahn), fin, 0, ws,
                                           wsn, st), "gte_gemm_f32 dh_neigh")
                aggregate(rcsr, w_out, t_out, dahn, fin, dh, fin, fin, _lib.REDUCE_SUM, True)

    def step(self, g, labels: torch.Tensor, n_global: Optional[int] = None,
             loss_scale: Optional[float] = None) -> torch.Tensor:
        scale = self._dp_scale(labels.shape[0], n_global, loss_scale)
        if self.distributed and self._dp_split:
            out3 = self.forward_backward(g, labels, scale, upto_layer=1)
            pending = [self._all_reduce_async(self.flat_grad[self._n0:])]    # layers 1.. : in flight under layer 0's backward
            self.backward_rest(g, 1)
            pending.append(self._all_reduce_async(self.flat_grad[:self._n0]))
            for w in pending:
                w.wait()
        elif self.distributed:
            out3 = self.forward_backward(g, labels, scale)
            import torch.distributed as dist
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        else:
            # one GPU: nothing sits between the gradient folds and Adam, so the optimiser state is brought up to date FIRST
            # and the step is applied by the fold launch itself (gte_fold_defer_flush_adam)
            self.t += 1
            try:
                self._sync_adam_state()
                out3 = self._forward_backward_adam(g, labels, scale)
            except BaseException:
                self.t -= 1
                raise
            return out3
        self.t += 1
        self._optimizer_step()
        return out3

    def _forward_backward_adam(self, g, labels, scale):
        """forward + backward + optimiser step with self.t already advanced and the device state synced: Adam inside the fold
        launch when the folds cover the whole gradient, its own launch otherwise."""
        self._fuse_adam_req, self._adam_fused = self.fuse_adam, False
        try:
            out3 = self.forward_backward(g, labels, scale)
        finally:
            self._fuse_adam_req = False
        if self._adam_fused:
            self._step_dev_host += 1
            self._adam_fused = False
            self.adam_fused_steps += 1
        else:
            self._adam_dev_launch()
        return out3

    def _all_reduce_async(self, t):
        import torch.distributed as dist
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # -- optimiser: hyper-parameters and step count live on the device, so the launch is graph-capturable ----------
    def _adam_host_state(self, t_next: int):
        """{lr, b1, b2, eps, wd, grad_scale, bc1, sqrt(bc2)} for step t_next, bias corrections in double (gte_adam_step)"""
        b1, b2 = float(self.betas[0]), float(self.betas[1])
        return (float(self.lr), b1, b2, float(self.eps), float(self.weight_decay), 1.0,
                float(np.float32(1.0 - b1 ** t_next)), float(np.float32(np.sqrt(1.0 - b2 ** t_next))))

    def _adam_state(self) -> None:
        if getattr(self, "_hyper", None) is None:
            dev = self.flat_param.device
            self._hyper_host = self._adam_host_state(1)
            self._hyper = torch.tensor(self._hyper_host, dtype=torch.float32, device=dev)
            self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            self._ticket = torch.zeros(int(self.lib.gte_adam_ticket_bytes()) // 4, dtype=torch.int32, device=dev)
            self._step_dev_host = 0                   # what the device counter holds (completed optimiser steps)

    def _sync_adam_state(self) -> None:
        """Eager-only: push changed hyper-parameters (lr_scale) / a changed step count (checkpoint restore) to the device.
        The device advances the counter and the bias corrections itself; the host only mirrors the expected values."""
        self._adam_state()
        want = self._adam_host_state(self.t)
        stale_t = self._step_dev_host != self.t - 1
        if stale_t or want[:6] != self._hyper_host[:6]:
            if stale_t:
                self._step_dev.fill_(self.t - 1)
                self._step_dev_host = self.t - 1
                self._hyper.copy_(torch.tensor(want, dtype=torch.float32))
            else:
                self._hyper[:6].copy_(torch.tensor(want[:6], dtype=torch.float32))
            self._hyper_host = want

    def _adam_dev_launch(self) -> None:
        P = _lib.ptr
        self._wimg_sig = None
        last = getattr(self, "_last_wimg", None)
        self._last_wimg = None
        if (last is not None and self.wimg_in_fold and 0 < last[1] <= 12 and not torch.cuda.is_current_stream_capturing()):
            # the optimiser launch behind the all-reduce also writes the weight images of the step that just ran (one launch instead
            # of Adam + a conversion launch in front of the next forward: the one-GPU step has both inside its fold launch)
            wrote = ctypes.c_int(0)
            _lib.check(self.lib.gte_adam_step_dev_images(P(self.flat_param), P(self.flat_grad), P(self.exp_avg), P(self.exp_avg_sq),
                                                         self.flat_param.numel(), P(self._hyper), P(self._step_dev), P(self._ticket),
                                                         last[0], last[1], ctypes.byref(wrote), _lib.current_stream()),
                       "gte_adam_step_dev_images")
            if wrote.value:
                self._wimg_sig = (self._param_sig(), last[2])
            self._step_dev_host += 1
            return
        _lib.check(self.lib.gte_adam_step_dev(P(self.flat_param), P(self.flat_grad), P(self.exp_avg), P(self.exp_avg_sq),
                                              self.flat_param.numel(), P(self._hyper), P(self._step_dev), P(self._ticket),
                                              _lib.current_stream()), "gte_adam_step_dev")
        self._step_dev_host += 1

    def _optimizer_step(self) -> None:              # called with self.t already advanced
        self._sync_adam_state()
        self._adam_dev_launch()

    # -- HIP graph capture of a step on a RESIDENT batch ------------------------------------------------
    def capture(self, g, labels: torch.Tensor, n_global: Optional[int] = None, loss_scale: Optional[float] = None):
        """Returns ``replay() -> out3``: forward+backward of this batch as one HIP-graph launch, followed
        by Adam -- inside the same graph on one GPU (gte_adam_step_dev reads lr and the step count from device
        memory), eagerly after the all-reduce when distributed.  The batch's tensors must stay alive and unchanged
        in place (resident pages): the cache entry holds a reference to the graph object, so its id cannot be
        reused while the captured graph exists; :meth:`release` drops a captured batch and its ~300 MB of buffers."""
        scale = self._dp_scale(labels.shape[0], n_global, loss_scale)
        key = id(g)
        self._private_key = key                       # this batch's buffers are private to its graph (never reallocated)
        try:
            replay = self._capture(g, labels, scale, key)
            self._graph_owner[key] = (g, labels)
            return replay
        finally:
            self._private_key = None

    def release(self, g=None) -> None:
        """Forget the HIP graph(s) and private buffers captured for ``g`` (all captured batches when None)."""
        keys = list(self._graphs) if g is None else [id(g)]
        for k in keys:
            self._graphs.pop(k, None)
            self._graph_bufs.pop(k, None)
            for kk in [q for q in self._graph_bufs if isinstance(q, tuple) and q[0] == k]:     # (general plans: (batch, layout) keys)
                self._graph_bufs.pop(kk, None)
            self._graph_owner.pop(k, None)

    def _capture(self, g, labels, scale, key):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up on the side stream: buffers, CSR caches, props
            for _ in range(2):
                self.forward_backward(g, labels, scale)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._adam_state()
        # GTE_DP_GRAPH_COLLECTIVE=1 (experimental, off): the RCCL all-reduce is captured INSIDE the step's HIP graph, so a
        # data-parallel step is one graph launch like the single-GPU step.  Verified with one rank only (1-GPU boxes).
        graph_coll = self.distributed and os.environ.get("GTE_DP_GRAPH_COLLECTIVE", "0") == "1"
        in_graph_adam = (not self.distributed) or graph_coll   # otherwise the all-reduce sits between backward and Adam, eagerly
        split = self.distributed and self._dp_split and not graph_coll
        graph = torch.cuda.CUDAGraph()
        graph_b = torch.cuda.CUDAGraph() if split else None
        with torch.cuda.graph(graph):
            if in_graph_adam and not graph_coll:
                out3 = self._forward_backward_adam(g, labels, scale)
            else:
                out3 = self.forward_backward(g, labels, scale, upto_layer=1 if split else 0)
            if graph_coll:
                import torch.distributed as dist
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            if in_graph_adam and not graph_coll:
                pass                                  # (applied below, fused into the fold launch when possible)
            elif in_graph_adam:
                self._adam_dev_launch()               # reads lr / step count from device memory at replay time
        if split:                                     # layer 0's backward: replayed while the upper slice is all-reduced
            with torch.cuda.graph(graph_b, pool=graph.pool()):
                self.backward_rest(g, 1)
        if in_graph_adam:
            self._step_dev_host -= 1                  # capturing did not run it

        def replay():
            self._wimg_sig = None                     # the graph updates the parameters; its own forward converts the images
            if in_graph_adam:
                self.t += 1
                self._sync_adam_state()               # no-op unless lr / t were changed from outside
                graph.replay()
                self._step_dev_host += 1
                return out3
            graph.replay()
            if split:
                pending = [self._all_reduce_async(self.flat_grad[self._n0:])]
                graph_b.replay()
                pending.append(self._all_reduce_async(self.flat_grad[:self._n0]))
                for w in pending:
                    w.wait()
            else:
                import torch.distributed as dist
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.t += 1
            self._optimizer_step()
            return out3
        self._graphs[key] = (graph, graph_b)
        return replay
