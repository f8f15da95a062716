"""Data-parallel sharding of page graphs over the GPUs of one node.

The reference is single-device (src/models/model_train.py:124-130); this is new functionality
required by BASELINE.json config 5.  Page graphs never share edges, so the path shards
embarrassingly by page; the one exchange per step is the gradient all-reduce in
``models/engine.TrainStep``.  One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL
over xGMI); the same code runs on "gloo" for the CPU tests.

Everything here is deterministic host logic shared by all ranks (same seed => same plan), so no
communication is needed to agree on who owns which page or on the global node count of a step.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str = "nccl", device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return dist


def epoch_permutation(n_pages: int, seed: int, epoch: int) -> np.ndarray:
    """Shared shuffle of the page indices (the sklearn.utils.shuffle of model_train.py:279)."""
    return np.random.default_rng([seed, epoch]).permutation(n_pages)


def plan_epoch(page_sizes: Sequence[int], batch_pages: int, world: int, seed: int = 42, epoch: int = 0,
               drop_last: bool = True) -> List[List[np.ndarray]]:
    """steps x ranks page-id arrays.  Every step takes ``batch_pages * world`` pages of the shared
    permutation; inside a step the pages are dealt to ranks in snake order by descending node
    count, so each rank gets exactly ``batch_pages`` pages and a near-equal node count (page sizes
    vary ~10x: balancing by count alone would leave ranks idle at the all-reduce).  The tail that
    does not fill a global batch is dropped, as the reference drops it (model_train.py:283)."""
    sizes = np.asarray(page_sizes)
    perm = epoch_permutation(len(sizes), seed, epoch)
    per_step = batch_pages * world
    n_steps = len(perm) // per_step if drop_last else -(-len(perm) // per_step)
    if n_steps == 0:
        # the reference's `range(len(train_graphs) // batch_size)` (model_train.py:283) is empty too, and it then trains on
        # nothing while still writing checkpoints; with several ranks the global batch is batch_pages * world pages, so a
        # dataset that fills one single-GPU batch may not fill one here -- say so instead of running zero steps per epoch
        raise ValueError(f"{len(sizes)} training pages do not fill one global batch of {batch_pages} pages x {world} rank(s); "
                         f"lower TRAINING.batch_size (the global batch is batch_size * world_size pages)")
    plan = []
    for s in range(n_steps):
        ids = perm[s * per_step:(s + 1) * per_step]
        order = ids[np.argsort(-sizes[ids], kind="stable")]
        ranks: List[List[int]] = [[] for _ in range(world)]
        for j, pid in enumerate(order):
            r = j % (2 * world)
            ranks[r if r < world else 2 * world - 1 - r].append(int(pid))
        plan.append([np.asarray(sorted(r), dtype=np.int64) for r in ranks])
    return plan


def step_node_counts(plan: List[List[np.ndarray]], page_sizes: Sequence[int]) -> np.ndarray:
    """[steps, ranks] node counts; row sums are the n_global of each step."""
    sizes = np.asarray(page_sizes)
    return np.array([[int(sizes[ids].sum()) for ids in step] for step in plan], dtype=np.int64)


def step_weight_sums(plan: List[List[np.ndarray]], page_weight_sums: Sequence[float]) -> np.ndarray:
    """[steps, ranks] sums of the per-node class weights w[y_i] (``page_weight_sums[p]`` = that sum over page p).
    With ``nn.CrossEntropyLoss(weight)`` (model_train.py:171) a rank's loss is sum_local(w nll) / sum_local(w); scaling it by
    sum_local(w) / sum_global(w) makes the all-reduced gradient the single-GPU gradient of sum_all(w nll) / sum_all(w)."""
    ws = np.asarray(page_weight_sums, dtype=np.float64)
    return np.array([[float(ws[ids].sum()) for ids in step] for step in plan], dtype=np.float64)
