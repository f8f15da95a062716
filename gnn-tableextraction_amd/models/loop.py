"""The batch loop of ``train()`` (src/models/model_train.py:283-332 of the reference): a DIFFERENT batch of page
graphs every step, built on the device from the resident dataset, one optimisation step on it.

The reference does, per step, ``dgl.batch(train_batch).to(device)`` on the host (:286-298) and then the step
(:320-332).  Here the pages live in HBM (``graph.ResidentPages``) and a step's batch is four launches of index / row
copies -- and those launches do not sit on the step's critical path:

  * the page lists of ALL steps of an epoch are known when the epoch starts (``distributed.plan_epoch``), so the
    per-step metadata (page ids, node / edge offsets of the block-diagonal union) is computed once per epoch with
    vectorised numpy and uploaded in ONE pinned host->device copy;
  * batches are written into ``depth`` reused buffer sets (no allocation inside the loop);
  * batch s+1 is assembled on a SIDE STREAM while step s runs: the assembly is HBM-bound index work, the step is
    MFMA-bound, so they share the chip; two events per step order buffer reuse (HIP streams + events instead of
    host synchronisation: the host never waits for the device inside an epoch).

``bench.py`` times exactly this loop (``run_steps``); ``models.model_train.train`` runs it every epoch.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from .. import graph as G


class BatchPipeline:
    def __init__(self, resident: G.ResidentPages, depth: int = 2, side_stream: Optional[bool] = None):
        if depth < 2:
            raise ValueError("BatchPipeline needs at least two buffer sets (one being read, one being written)")
        self.res, self.depth = resident, depth
        self.device = resident.device
        # Where a batch is assembled.  "side": on the pipeline's own stream, under the kernels of the step before (events order
        # the buffer sets) -- what a batch that COPIES its feature rows needs (~45 us for 100 pages x 831 fp32 columns).  "same":
        # in the caller's stream order, no events at all -- a row-map batch (image mode: page table, CSRs and labels only) is
        # 7 us there against 14-20 us beside a GEMM plus three event packets on the caller's stream per step (profiles/r05/
        # side_stream_ab.txt: -7 ... -12 us per step on every shape).  Decided per load(): GTE_PIPE_SIDE=1 / 0 force either.
        self.side = torch.cuda.Stream(device=self.device)
        self._force_side = {"1": True, "0": False}.get(os.environ.get("GTE_PIPE_SIDE", ""), side_stream)     # None: per load()
        self._same = False
        self._meta_pending = False                      # the caller's stream has not yet waited for the last metadata upload
        self._ride = os.environ.get("GTE_PIPE_RIDE", "1") == "1"       # (0: measurement -- the in-stream assembly as its own launch)
        from .. import _lib as _gte_lib
        self._lib = _gte_lib.load()
        self._sets: List[dict] = []
        self._free_ev: List[Optional[torch.cuda.Event]] = [None] * depth
        self._pinned = None
        self._meta_dev = None
        self._meta_ev = None
        self._info = None
        self._ready = {}
        # per-page entry counts of both CSRs, host side (numpy): the offsets of any batch are cumulative sums of these
        self._page_nodes = (resident.node_off_host[1:] - resident.node_off_host[:-1]).numpy().astype(np.int64)
        self._page_ent = []
        for name in ("in", "out"):
            eo = resident._sets[name]["edge_off_host"].numpy().astype(np.int64)
            self._page_ent.append(eo[1:] - eo[:-1])

    def rebind(self, resident: G.ResidentPages) -> None:
        """Another resident set of the same layout (the next WINDOW of a host-resident dataset, models/residency.py): the batch
        buffer sets stay, the page table behind the metadata changes.  The caller orders the streams (every step on the old
        set has been queued; its buffers are protected by the per-set events as before)."""
        self.res = resident
        self._ready = {}
        self._info = None
        self._page_nodes = (resident.node_off_host[1:] - resident.node_off_host[:-1]).numpy().astype(np.int64)
        self._page_ent = []
        for name in ("in", "out"):
            eo = resident._sets[name]["edge_off_host"].numpy().astype(np.int64)
            self._page_ent.append(eo[1:] - eo[:-1])
        self._bound_nb = getattr(self, "_bound_nb", None) if getattr(self, "_bound_pages", None) else None

    # ---- per epoch -------------------------------------------------------------------------------
    def load(self, steps: Sequence[np.ndarray]) -> None:
        """Metadata of every step of an epoch (``steps[s]`` = page ids of step s): one upload."""
        self._ready = {}
        self._info = []
        if len(steps) == 0:
            return
        rows = []
        for ids in steps:
            ids = np.asarray(ids, dtype=np.int64)
            if ids.size == 0:
                raise ValueError("a step needs at least one page")
            b_node = np.zeros(ids.size + 1, dtype=np.int64)
            np.cumsum(self._page_nodes[ids], out=b_node[1:])
            b_in = np.zeros(ids.size + 1, dtype=np.int64)
            np.cumsum(self._page_ent[0][ids], out=b_in[1:])
            b_out = np.zeros(ids.size + 1, dtype=np.int64)
            np.cumsum(self._page_ent[1][ids], out=b_out[1:])
            rows.append((ids, b_node, b_in, b_out))
        total = sum(4 * r[0].size + 3 for r in rows)
        # two pinned staging areas, used alternately: the upload of the PREVIOUS load() may still be queued behind that load's
        # batch assemblies (which wait for the steps that free their buffers) -- waiting for it here would stall the host once per
        # load, i.e. once per window chunk of a host-resident dataset (models/residency.py); the one before that is long done
        if getattr(self, "_pin2", None) is None:
            self._pin2, self._pin_ev, self._pin_k = [None, None], [None, None], 0
        self._pin_k ^= 1
        k = self._pin_k
        if self._pin_ev[k] is not None:
            self._pin_ev[k].synchronize()
        if self._pin2[k] is None or self._pin2[k].numel() < total:
            self._pin2[k] = torch.empty(max(total, 4096), dtype=torch.int32).pin_memory()
        self._pinned = self._pin2[k]
        # ... and two DEVICE tables, used alternately as well: the upload of this load() goes into the table the load() before the
        # previous one used, whose assemblies were queued a whole load() ago -- it waits (on the side stream) for an event that has
        # long passed instead of for everything queued on the caller's stream up to now (round 5: one full host-to-device latency
        # per window chunk of a host-resident set, in front of the next step)
        if getattr(self, "_meta2", None) is None:
            self._meta2, self._meta_done = [None, None], [None, None]
        cur = torch.cuda.current_stream(self.device)
        if self._meta_dev is not None and self._meta2[k ^ 1] is self._meta_dev:
            # every assembly that reads the previous load()'s table has been queued by now: the mark its next overwrite waits for
            self._meta_done[k ^ 1] = torch.cuda.Event()
            self._meta_done[k ^ 1].record(cur)
        if self._meta2[k] is None or self._meta2[k].numel() < total:
            if self._meta_done[k] is not None:
                self._meta_done[k].synchronize()                 # (a larger device table: the old one may still be read)
            if self._meta_ev is not None:
                self._meta_ev.synchronize()
            self._meta2[k] = torch.empty(max(self._pinned.numel(), total), dtype=torch.int32, device=self.device)
            self._meta2[k].record_stream(self.side)              # (read on the side stream too: no reuse under it once dropped)
            self._meta_done[k] = None
        self._meta_dev = self._meta2[k]
        stage = self._pinned.numpy()
        off = 0
        cap = [0, 0, 0]
        for ids, b_node, b_in, b_out in rows:
            nb = ids.size
            stage[off:off + nb] = ids
            stage[off + nb:off + 2 * nb + 1] = b_node
            stage[off + 2 * nb + 1:off + 3 * nb + 2] = b_in
            stage[off + 3 * nb + 2:off + 4 * nb + 3] = b_out
            self._info.append((off, nb, int(b_node[-1]), int(b_in[-1]), int(b_out[-1]), self._page_nodes[ids]))
            cap = [max(cap[0], int(b_node[-1])), max(cap[1], int(b_in[-1])), max(cap[2], int(b_out[-1]))]
            off += 4 * nb + 3
        same = (self.res.p3_mode == "rows") if self._force_side is None else (not self._force_side)
        if same != self._same and self._meta_ev is not None:
            torch.cuda.synchronize(self.device)         # (the resident pages changed mode: rare; no ordering left to think about)
            self._free_ev = [None] * self.depth
        self._same = same
        # the side stream runs in order: assemblies queued there earlier read their table before a later copy overwrites it; those
        # queued on the caller's stream are covered by the mark recorded one load() ago (above)
        if self._meta_done[k] is not None:
            self.side.wait_event(self._meta_done[k])
        self._meta_pending = self._same                  # (side mode: the assemblies run behind the upload on the side stream itself)
        with torch.cuda.stream(self.side):
            self._meta_dev[:total].copy_(self._pinned[:total], non_blocking=True)
            self._meta_ev = torch.cuda.Event()
            self._meta_ev.record(self.side)
            self._pin_ev[self._pin_k] = self._meta_ev
        # Capacity for ANY batch of this many pages (the nb largest pages of the dataset), not just this epoch's largest: a
        # later epoch with a bigger batch would otherwise reallocate (synchronise + allocate ~1 ms) in the middle of the loop.
        nb_max = max(r[0].size for r in rows)
        if nb_max != getattr(self, "_bound_nb", None):
            top = lambda a: int(np.sort(a)[-nb_max:].sum())
            # (a windowed dataset bounds the batch by the pages of the WHOLE dataset: no reallocation when the window changes)
            src = getattr(self, "_bound_pages", None) or (self._page_nodes, self._page_ent[0], self._page_ent[1])
            self._bound = [top(src[0]), top(src[1]), top(src[2])]
            self._bound_nb = nb_max
        self._ensure_capacity([max(c, b) for c, b in zip(cap, self._bound)])

    def max_batch_nodes(self) -> int:
        """Upper bound of the node count of any batch the loaded plan's batch size can produce."""
        return int(self._bound[0]) if getattr(self, "_bound", None) else 0

    def _ensure_capacity(self, cap) -> None:
        have = self._sets[0]["cap"] if self._sets else (0, 0, 0)
        if self._sets and all(h >= c for h, c in zip(have, cap)):
            return
        torch.cuda.synchronize(self.device)                      # rare (first epoch / a larger batch than any before)
        grow = [max(h, -(-int(c * 1.0625) // 1024) * 1024) for h, c in zip(have, cap)]
        self._sets = [self.res.alloc_batch_buffers(*grow) for _ in range(self.depth)]
        self._free_ev = [None] * self.depth

    def __len__(self):
        return len(self._info or [])

    def nodes(self, s: int) -> int:
        return self._info[s][2]

    # ---- per step --------------------------------------------------------------------------------
    def start(self, s: int, after_current: bool = False) -> None:
        """Queue the assembly of step s's batch (returns immediately): in the caller's stream order or on the side stream (see
        __init__; decided by the last load()).  ``after_current``: the caller is inside a step, in front of its last GEMM -- a
        side-stream assembly additionally waits for everything queued on the CURRENT stream so far (which places it under that
        GEMM), an in-stream assembly rides the step's fold + optimiser launch."""
        off, nb, n, e_in, e_out, n_sizes = self._info[s]
        k = s % self.depth
        if self._same and self._meta_pending:
            torch.cuda.current_stream(self.device).wait_event(self._meta_ev)
            self._meta_pending = False
        if self._free_ev[k] is not None and not self._same:
            self.side.wait_event(self._free_ev[k])               # the step that last read this buffer set has finished
        if after_current and not self._same and self.side is not torch.cuda.current_stream(self.device):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.side.wait_event(ev)
        # in stream order and from inside a step (the engine's before-last-GEMM hook): the assembly rides the step's fold +
        # optimiser launch as extra workgroups (gte_batch_assemble_defer) -- it writes the OTHER buffer set, which nothing of
        # this step reads; outside a step (no fold deferral open) the same call launches at once
        ride = self._same and after_current and self._ride
        if ride:
            self._lib.gte_batch_assemble_defer(1)
        try:
            g = self.res.assemble(self._meta_dev[off:off + 4 * nb + 3], nb, n, e_in, e_out, self._sets[k], n_sizes,
                                  stream=(torch.cuda.current_stream(self.device) if self._same else self.side).cuda_stream)
        finally:
            if ride:
                self._lib.gte_batch_assemble_defer(0)
        ev = None
        if not self._same:
            ev = torch.cuda.Event()
            ev.record(self.side)
        self._ready[s] = (g, ev)

    def get(self, s: int) -> G.ResidentBatch:
        """The batch of step s; the CURRENT stream waits (on the device) for its assembly."""
        g, ev = self._ready.pop(s)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return g

    def release(self, s: int) -> None:
        """Step s has been queued on the current stream: its buffer set may be rewritten once it is through."""
        if self._same:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._free_ev[s % self.depth] = ev


def run_steps(step, pipe: BatchPipeline, page_steps: Sequence[np.ndarray], n_global: Optional[Sequence[int]] = None,
              loss_scale: Optional[Sequence[float]] = None, on_step: Optional[Callable] = None):
    """One optimisation step per entry of ``page_steps`` (this rank's page ids per step), batches assembled one step
    ahead on the pipeline's side stream.  ``n_global[s]`` = node count of step s over all ranks, ``loss_scale[s]`` an
    explicit local loss factor (class-weighted data parallelism).  Returns the last step's device vector
    [loss, sum of class weights, #correct] (not synchronised)."""
    want_p3 = bool(step.wants_resident_images(pipe.res.feat.shape[1]) if hasattr(step, "wants_resident_images") else
                   (hasattr(step, "wants_p3_features") and step.wants_p3_features(pipe.res.feat.shape[1])))
    if want_p3 != bool(pipe.res.p3_mode):
        # layer 0 multiplies a P3 image (planes GEMMs): resident features and batches become images -- or back to fp32 rows when
        # the GEMM mode was switched; the buffer sets of the other kind are dropped
        torch.cuda.synchronize(pipe.device)
        # (agg: layer 0 on the cached mean aggregate of the input -- a second resident image, engine.wants_agg_image)
        pipe.res.enable_p3(agg=bool(getattr(step, "wants_agg_image", lambda f: False)(pipe.res.feat.shape[1]))) if want_p3 else pipe.res.disable_p3()
        pipe._sets = []
        pipe._free_ev = [None] * pipe.depth
    pipe.load(page_steps)
    n_steps = len(pipe)
    out3 = None
    if n_steps == 0:
        return out3
    if hasattr(step, "reserve"):                 # the engine's per-batch buffers: sized once for the largest possible batch
        step.reserve(pipe.max_batch_nodes(), pipe.res.feat.shape[1],
                     cached=bool(pipe.res.p3_mode == "rows" and pipe.res.agg_p3 is not None))
    pipe.start(0)
    # where the next batch is assembled: an engine with a `before_last_gemm` hook (FusedGcnSageStep) gets it under the last,
    # MFMA-bound GEMM of the current step -- late enough that the batch is still cache-resident when the next step starts
    # reading it, early enough to be off the critical path; other engines start it before the step
    hooked = hasattr(step, "before_last_gemm") and os.environ.get("GTE_PIPE_LATE", "1") == "1"
    pending = [None]

    def late_start():
        if pending[0] is not None:
            pipe.start(pending[0], after_current=True)
            pending[0] = None
    if hooked:
        step.before_last_gemm = late_start
    try:
        for s in range(n_steps):
            if s + 1 < n_steps:
                if hooked:
                    pending[0] = s + 1
                else:
                    pipe.start(s + 1)
            g = pipe.get(s)
            out3 = step.step(g, g.ndata['label'], n_global=None if n_global is None else int(n_global[s]),
                             loss_scale=None if loss_scale is None else float(loss_scale[s]))
            late_start()                                 # engines / layouts that never reached the hook
            pipe.release(s)
            if on_step is not None:
                on_step(s, g, out3)
    finally:
        if hooked:
            step.before_last_gemm = None
    return out3
