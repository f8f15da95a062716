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

from typing import Optional

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

    def step(self, g, labels: torch.Tensor, n_global: Optional[int] = None) -> torch.Tensor:
        """Forward, loss, backward, (all-reduce), Adam.  Returns the device vector
        [loss (local mean), sum of class weights, #correct] without synchronising."""
        self.model.train()
        self.flat_grad.zero_()
        logits = self.model(g)
        loss, out3 = self._loss(logits, labels)
        n_local = labels.shape[0]
        if self.distributed and n_global:
            loss = loss * (float(n_local) / float(n_global))
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
