"""Training sets larger than the HBM budget: the pages live in pinned HOST memory, a WINDOW of them is resident.

The reference streams every batch from the host (src/models/model_train.py:283-298: slice the page list, build features on the
CPU, ``dgl.batch(...).to(device)``).  ``models/model_train.train`` keeps the whole training set in HBM (graph.ResidentPages) --
which stops at 288 GB per GPU: PubLayNet's full train split at F0 = 831 is ~0.5 TB of fp32 features.  This module is the tier
above it (``GTE_RESIDENT_BUDGET_GB``):

* **HostPages** -- this rank's pages only (data parallelism: a fixed page -> rank ownership, so the residency shrinks with the
  world size), concatenated once in pinned host memory in the array layout of graph.ResidentPages (features, labels, per-page
  local CSRs of both directions, CSR-ordered weights): a contiguous page range is a handful of contiguous slices.
* **WindowedPages** -- two device slots of budget / 2 each; a window = a contiguous page range that fits a slot.  While the
  steps of window k run, window k + 1 is uploaded on a copy stream (contiguous pinned slices -> cudaMemcpyAsync at PCIe rate,
  no host gather), converted to the P3 image there when layer 0 takes one, and handed over through events.  Windows stay below
  4 GB of image, so the input layer keeps reading the resident image through the row map.
* **WindowStream** -- the order of steps.  A step of the headline configuration consumes 81 MB of features in 0.55 ms = 147 GB/s,
  more than twice what PCIe Gen5 x16 delivers: a loop that uploads every page once per epoch CANNOT run at the all-resident
  rate.  So a window is visited for ``passes`` shuffled passes before the next one is taken (every page is still seen once per
  epoch-equivalent of steps on average; within a burst of ``passes`` epochs' worth it is seen ``passes`` times in a row, then
  not until the stream returns to its window).  An epoch stays what it is in the reference -- len(train) // batch_size steps,
  then validation -- and simply takes the next steps of the stream.  Deterministic host logic: every rank can compute every
  rank's stream (node counts of a step without communication, like distributed.plan_epoch).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import graph as G
from .. import ops


def page_owner(n_pages: int, world: int, seed: int = 42) -> np.ndarray:
    """rank that owns each page: a seeded shuffle dealt round-robin (equal page counts, sizes mixed)."""
    perm = np.random.default_rng([seed, 977]).permutation(n_pages)
    owner = np.empty(n_pages, dtype=np.int64)
    owner[perm] = np.arange(n_pages) % max(world, 1)
    return owner


def window_page_order(n_pages: int, seed: int = 42, rank: int = 0) -> np.ndarray:
    """The order in which a rank lays its pages out in host memory: a seeded permutation.  Windows are CONTIGUOUS ranges of that
    layout; in dataset order (papers, or -- the adversarial case -- pages sorted by size) a window is a biased sample of the set and a
    loop that trains on one window at a time ends biased towards the windows it saw last: measured 0.79 against 0.34 validation
    loss after 1 500 steps on size-sorted windows (tests/test_gpu_residency.py).  Shuffled once, every window is an unbiased sample."""
    return np.random.default_rng([seed, 1213, rank]).permutation(n_pages)


def window_ranges(page_nodes: np.ndarray, bytes_per_node: float, slot_bytes: float, max_nodes: Optional[int] = None) -> List[Tuple[int, int]]:
    """Contiguous page ranges [p0, p1) whose nodes fit a device slot (and ``max_nodes``: the 4 GB bound of the row map)."""
    cap = int(slot_bytes // bytes_per_node)
    if max_nodes is not None:
        cap = min(cap, int(max_nodes))
    big = int(np.max(page_nodes)) if len(page_nodes) else 0
    if big > cap:
        raise ValueError(f"a page with {big} nodes does not fit a window of {cap} nodes: raise GTE_RESIDENT_BUDGET_GB")

    def greedy(c):
        out, p0, acc = [], 0, 0
        for p, n in enumerate(page_nodes):
            if acc + n > c:
                out.append((p0, p))
                p0, acc = p, 0
            acc += int(n)
        out.append((p0, len(page_nodes)))
        return out
    out = greedy(cap)
    # the same NUMBER of windows, balanced: the smallest capacity that still needs no more windows.  (Filling every window to the
    # brim leaves a last range of whatever remains -- possibly fewer pages than one batch, which the stream would then never
    # train on.)
    lo, hi = max(big, int(np.sum(page_nodes)) // max(len(out), 1)), cap
    while lo < hi:
        mid = (lo + hi) // 2
        if len(greedy(mid)) <= len(out):
            hi = mid
        else:
            lo = mid + 1
    return greedy(hi)


class WindowStream:
    """The endless sequence of steps of ONE rank: windows in a seeded order per sweep, ``passes`` shuffled passes over a window
    per visit, ``batch_pages`` pages per step (the tail of a pass is dropped, as the reference drops an epoch's tail).
    ``take(n)`` -> [(window index, [page ids local to the window, one array per step])] for the next n steps."""

    def __init__(self, ranges: Sequence[Tuple[int, int]], batch_pages: int, passes: int, seed: int, rank: int = 0):
        self.ranges, self.B, self.passes, self.seed, self.rank = list(ranges), int(batch_pages), max(int(passes), 1), seed, rank
        if all(p1 - p0 < self.B for p0, p1 in self.ranges):
            raise ValueError(f"no window holds {self.B} pages: raise GTE_RESIDENT_BUDGET_GB or lower TRAINING.batch_size")
        self.sweep, self.pos, self.pas, self.off = 0, 0, 0, 0          # sweep, position in its window order, pass, step in pass
        self._order = self._window_order(0)
        self._perm = None

    def never_visited(self) -> List[int]:
        """pages (local ids) of windows that hold fewer than one batch: the stream skips such a window, so they are never trained
        on (balanced window_ranges make this a corner case -- a rank with fewer pages than two batches; callers report it)"""
        return [p for p0, p1 in self.ranges if p1 - p0 < self.B for p in range(p0, p1)]

    def skip(self, n_steps: int) -> None:
        """Advance by n_steps without producing them: a resumed run (checkpoint at epoch e) continues the stream where the
        interrupted run stood -- the position is a function of the number of steps taken, like distributed.plan_epoch's plan
        is a function of the epoch."""
        while n_steps > 0:
            w = int(self._order[self.pos])
            p0, p1 = self.ranges[w]
            per_pass = (p1 - p0) // self.B
            if per_pass == 0:
                self._advance_window()
                continue
            left_in_visit = (self.passes - self.pas) * per_pass - self.off
            if n_steps >= left_in_visit:                          # whole rest of this window visit
                n_steps -= left_in_visit
                self.off, self.pas, self._perm = 0, 0, None
                self._advance_window()
            else:
                done = self.pas * per_pass + self.off + n_steps
                self.pas, self.off, self._perm = done // per_pass, done % per_pass, None
                n_steps = 0

    def _window_order(self, sweep):
        return np.random.default_rng([self.seed, 31, self.rank, sweep]).permutation(len(self.ranges))

    def _pass_perm(self, w):
        p0, p1 = self.ranges[w]
        return np.random.default_rng([self.seed, 57, self.rank, self.sweep, int(w), self.pas]).permutation(p1 - p0)

    def peek_window(self) -> int:
        return int(self._order[self.pos])

    def next_window(self) -> int:
        """the window after the current one (what to upload while the current one trains)"""
        if self.pos + 1 < len(self._order):
            return int(self._order[self.pos + 1])
        return int(self._window_order(self.sweep + 1)[0])

    def take(self, n_steps: int):
        out = []
        while n_steps > 0:
            w = int(self._order[self.pos])
            p0, p1 = self.ranges[w]
            per_pass = (p1 - p0) // self.B
            if per_pass == 0:                                   # a window too small for one batch (the last range): skip it
                self._advance_window()
                continue
            if self._perm is None:
                self._perm = self._pass_perm(w)
            k = min(n_steps, per_pass - self.off)
            steps = [np.sort(self._perm[(self.off + i) * self.B:(self.off + i + 1) * self.B]) for i in range(k)]
            if out and out[-1][0] == w:
                out[-1][1].extend(steps)
            else:
                out.append((w, steps))
            self.off += k
            n_steps -= k
            if self.off == per_pass:
                self.off, self._perm = 0, None
                self.pas += 1
                if self.pas == self.passes:
                    self.pas = 0
                    self._advance_window()
        return out

    def _advance_window(self):
        self.pos += 1
        if self.pos == len(self._order):
            self.sweep += 1
            self.pos = 0
            self._order = self._window_order(self.sweep)


class HostPages:
    """This rank's pages in pinned host memory, in the layout of graph.ResidentPages (built chunk by chunk through it)."""

    def __init__(self, graphs: Sequence[G.PageGraph], device, chunk_bytes: int = 1 << 30, pin: bool = True):
        self.device = torch.device(device)
        sizes = np.array([g.num_nodes() for g in graphs], dtype=np.int64)
        edges = np.array([g.num_edges() for g in graphs], dtype=np.int64)
        self.n_pages = len(graphs)
        F = int(graphs[0].ndata['feat'].shape[1])
        self.n_feat = F
        N, E, P = int(sizes.sum()), int(edges.sum()), len(graphs)
        self.node_off = np.zeros(P + 1, dtype=np.int64)
        np.cumsum(sizes, out=self.node_off[1:])
        self.page_nodes, self.page_edges = sizes, edges

        def host(shape, dtype):
            t = torch.empty(shape, dtype=dtype)
            return t.pin_memory() if pin else t
        self.feat = host((N, F), torch.float32)
        self.has_label = graphs[0].ndata.get('label') is not None
        self.label = host((N, 1), torch.float32) if self.has_label else None
        self.weighted = graphs[0].edata.get('feat') is not None
        self.sets = {name: dict(indptr_loc=host((N + P,), torch.int32), indices_loc=host((max(E, 1),), torch.int32),
                                weight=None, edge_off=np.zeros(P + 1, dtype=np.int64)) for name in ("in", "out")}
        self.max_deg = {"in": 0, "out": 0}
        p0 = 0
        while p0 < P:
            p1, acc = p0, 0
            while p1 < P and (p1 == p0 or acc + sizes[p1] * F * 4 <= chunk_bytes):
                acc += sizes[p1] * F * 4
                p1 += 1
            # (the CSR arrays of the chunk are built on the device; the feature rows go from the page tensors straight into the
            # pinned matrix -- through the device they cost 99 s for 60 000 pages)
            rp = G.ResidentPages(graphs[p0:p1], self.device, with_feat=False)
            n0, n1 = int(self.node_off[p0]), int(self.node_off[p1])
            G.upload_rows([g.ndata['feat'] for g in graphs[p0:p1]], None, out=self.feat[n0:n1])
            if self.has_label:
                self.label[n0:n1].copy_(rp.label)
            for name in ("in", "out"):
                st, mine = rp._sets[name], self.sets[name]
                eo = st["edge_off_host"].numpy().astype(np.int64)
                base = mine["edge_off"][p0]
                mine["edge_off"][p0:p1 + 1] = base + eo
                mine["indices_loc"][base:base + eo[-1]].copy_(st["indices_loc"])
                mine["indptr_loc"][n0 + p0:n1 + p1].copy_(st["indptr_loc"])
                if st["weight"] is not None:
                    if mine["weight"] is None:
                        mine["weight"] = host((max(E, 1),), torch.float32)
                    mine["weight"][base:base + eo[-1]].copy_(st["weight"])
                self.max_deg[name] = max(self.max_deg[name], rp.max_deg[name])
            del rp
            p0 = p1
        torch.cuda.synchronize(self.device)

    def feature_bytes(self) -> int:
        return int(self.feat.numel()) * 4


class WindowedPages:
    """Two device slots; ``acquire(w)`` hands out window w as a graph.ResidentPages (its upload was started by ``prefetch``)."""

    ROW_MAP_LIMIT = (1 << 32) - (1 << 20)

    @staticmethod
    def bytes_per_node(page_nodes: np.ndarray, page_edges: np.ndarray, n_feat: int, want_p3: bool, want_agg: bool = False) -> float:
        """HBM bytes of a resident node: its feature row (P3 image, or fp32; ``want_agg``: and the image of its input's mean
        aggregate) + label + both CSRs (indptr, indices, weights)."""
        ldp = 96 * (-(-n_feat // 16)) if want_p3 else 0
        deg = max(float(page_edges.sum()) / max(float(page_nodes.sum()), 1.0), 1.0)
        return (ldp * (2 if want_agg else 1) if want_p3 else 4 * n_feat) + 4 + 2 * (4 + deg * 8) + 8

    @staticmethod
    def layout(page_nodes: np.ndarray, page_edges: np.ndarray, n_feat: int, budget_bytes: float, want_p3: bool,
               want_agg: bool = False) -> List[Tuple[int, int]]:
        """The window page ranges of a rank: pure host arithmetic on the page table (every rank can compute every rank's)."""
        ldp = 96 * (-(-n_feat // 16)) if want_p3 else 0
        per_node = WindowedPages.bytes_per_node(page_nodes, page_edges, n_feat, want_p3, want_agg)
        # (the fp32 staging rows of the image conversion are shared by the two slots: one upload at a time)
        stage = 4 * n_feat if want_p3 else 0
        slot_bytes = budget_bytes / (2.0 + stage / per_node)
        max_nodes = (WindowedPages.ROW_MAP_LIMIT // ldp) if want_p3 else None
        return window_ranges(page_nodes, per_node, slot_bytes, max_nodes)

    def __init__(self, host: HostPages, budget_bytes: float, want_p3: bool, want_agg: bool = False):
        """``want_agg`` (with ``want_p3``): every window also carries the image of the input's mean aggregate, computed on the copy
        stream from the uploaded rows and the window's own CSR (graph.ResidentPages.build_agg_image): the input layer then runs
        on the cached aggregate (engine GTE_LAYER_CACHED) exactly as on an all-resident set."""
        self.host, self.device, self.want_p3 = host, host.device, bool(want_p3)
        self.want_agg = bool(want_agg and want_p3)
        F = host.n_feat
        self.ldp = 96 * (-(-F // 16)) if want_p3 else 0
        self.ranges = self.layout(host.page_nodes, host.page_edges, F, budget_bytes, want_p3, self.want_agg)
        dev = self.device
        nmax = max(int(host.node_off[p1] - host.node_off[p0]) for p0, p1 in self.ranges)
        pmax = max(p1 - p0 for p0, p1 in self.ranges)
        emax = max(int(host.sets["in"]["edge_off"][p1] - host.sets["in"]["edge_off"][p0]) for p0, p1 in self.ranges)
        emax = max(emax, max(int(host.sets["out"]["edge_off"][p1] - host.sets["out"]["edge_off"][p0]) for p0, p1 in self.ranges), 1)
        self.stage = torch.empty((nmax, F), dtype=torch.float32, device=dev) if want_p3 else None
        self.slots = []
        for _ in range(2):
            sl = {"feat": torch.empty((nmax, self.ldp), dtype=torch.uint8, device=dev) if want_p3
                  else torch.empty((nmax, F), dtype=torch.float32, device=dev),
                  "agg": torch.empty((nmax, self.ldp), dtype=torch.uint8, device=dev) if self.want_agg else None,
                  "label": torch.empty((nmax, 1), dtype=torch.float32, device=dev) if host.has_label else None,
                  "window": None, "ready": None, "free": None, "res": None,
                  # page tables of the window (node / in-edge / out-edge offsets): pinned staging + device copy, so that their
                  # upload is asynchronous like the big slices (a pageable copy blocks the host until the copy stream has
                  # drained -- i.e. until the steps of the slot's previous window are through: a bubble per window)
                  "meta_host": torch.empty(3 * (pmax + 1), dtype=torch.int32).pin_memory(),
                  "meta_dev": torch.empty(3 * (pmax + 1), dtype=torch.int32, device=dev), "meta_ev": None}
            for name in ("in", "out"):
                sl[name] = dict(indptr_loc=torch.empty(nmax + pmax, dtype=torch.int32, device=dev),
                                indices_loc=torch.empty(emax, dtype=torch.int32, device=dev),
                                weight=torch.empty(emax, dtype=torch.float32, device=dev) if host.sets[name]["weight"] is not None else None)
            self.slots.append(sl)
        # A stream of its own.  Measured alternatives (profiles/r04/residency.md): the upload pieced out on the batch pipeline's
        # side stream behind every assembly (0.52 / 0.68 of the all-resident rate at 4 / 8 passes: a piece in front of an assembly
        # that waits for the step before it stalls the next piece), a high-priority stream (0.42 / 0.48), more HIP hardware
        # queues (GPU_MAX_HW_QUEUES 8 ... 32: 0.53 / 0.98).  What the uploads get while the step's kernels run is 21 ... 38 GB/s of
        # the link's 57 (host memory and PCIe are shared with the other GPUs' jobs of the node: it differs run to run).
        self.copy = torch.cuda.Stream(device=dev)
        self.delay_cycles = 0                                 # test hook: spin this many GPU cycles in front of every upload
        self._no_feat = torch.empty((0, F), dtype=torch.float32, device=dev)
        self.device_bytes = sum(t.numel() * t.element_size() for sl in self.slots for t in
                                [sl["feat"], sl["agg"], sl["label"]] + [v for n in ("in", "out") for v in sl[n].values()] if t is not None)
        self.device_bytes += self.stage.numel() * 4 if self.stage is not None else 0
        self.uploaded_bytes = 0

    def _slot_of(self, w):
        for sl in self.slots:
            if sl["window"] == w:
                return sl
        return None

    def prefetch(self, w: int) -> None:
        """Start the upload of window w into the slot that does not hold the window in use (no-op if it is resident)."""
        if self._slot_of(w) is not None:
            return
        sl = next(s for s in self.slots if s.get("in_use") is not True)
        h = self.host
        p0, p1 = self.ranges[w]
        n0, n1 = int(h.node_off[p0]), int(h.node_off[p1])
        n = n1 - n0
        np_ = p1 - p0
        if sl["meta_ev"] is not None:
            sl["meta_ev"].synchronize()                       # (the previous upload from this pinned staging area has left the host)
        mh = sl["meta_host"].numpy()
        mh[:np_ + 1] = h.node_off[p0:p1 + 1] - n0
        for k, name in enumerate(("in", "out")):
            eo_ = h.sets[name]["edge_off"]
            mh[(k + 1) * (np_ + 1):(k + 2) * (np_ + 1)] = eo_[p0:p1 + 1] - eo_[p0]
        with torch.cuda.stream(self.copy):
            if sl["free"] is not None:
                self.copy.wait_event(sl["free"])              # the last step that read this slot's old window has run
            if self.delay_cycles:
                torch.cuda._sleep(int(self.delay_cycles))     # (tests: an upload that is late -- every reader must wait for it)
            sl["meta_dev"][:3 * (np_ + 1)].copy_(sl["meta_host"][:3 * (np_ + 1)], non_blocking=True)
            sl["meta_ev"] = torch.cuda.Event()
            sl["meta_ev"].record(self.copy)
            sets = {}
            for name in ("in", "out"):
                hs, ds = h.sets[name], sl[name]
                b0, b1 = int(hs["edge_off"][p0]), int(hs["edge_off"][p1])
                ds["indices_loc"][:b1 - b0].copy_(hs["indices_loc"][b0:b1], non_blocking=True)
                ds["indptr_loc"][:n + (p1 - p0)].copy_(hs["indptr_loc"][n0 + p0:n1 + p1], non_blocking=True)
                wt = None
                if ds["weight"] is not None:
                    ds["weight"][:b1 - b0].copy_(hs["weight"][b0:b1], non_blocking=True)
                    wt = ds["weight"][:b1 - b0]
                eo = torch.from_numpy(hs["edge_off"][p0:p1 + 1] - b0)
                kk = 1 if name == "in" else 2
                sets[name] = dict(edge_off=sl["meta_dev"][kk * (np_ + 1):(kk + 1) * (np_ + 1)], edge_off_host=eo,
                                  indices_loc=ds["indices_loc"][:max(b1 - b0, 1)], indptr_loc=ds["indptr_loc"][:n + (p1 - p0)], weight=wt)
            label = None
            if sl["label"] is not None:
                sl["label"][:n].copy_(h.label[n0:n1], non_blocking=True)
                label = sl["label"][:n]
            node_off = torch.from_numpy(h.node_off[p0:p1 + 1] - n0)
            if self.want_p3:
                self.stage[:n].copy_(h.feat[n0:n1], non_blocking=True)
                img = ops.P3(sl["feat"], n, h.n_feat)
                ops.p3_from_f32(self.stage[:n], out=img)
                feat = self._no_feat
                res = G.ResidentPages.from_arrays(self.device, node_off, feat, label, sets, h.weighted, h.max_deg,
                                                  feat_p3=ops.P3(sl["feat"][:n], n, h.n_feat), p3_mode="rows",
                                                  node_off_dev=sl["meta_dev"][:np_ + 1])
                if self.want_agg:                             # norm . A_w x of the window's rows, from the staged fp32 rows
                    res.build_agg_image(x_f32=self.stage[:n], out=ops.P3(sl["agg"][:n], n, h.n_feat))
            else:
                sl["feat"][:n].copy_(h.feat[n0:n1], non_blocking=True)
                res = G.ResidentPages.from_arrays(self.device, node_off, sl["feat"][:n], label, sets, h.weighted, h.max_deg,
                                                  node_off_dev=sl["meta_dev"][:np_ + 1])
            ev = torch.cuda.Event()
            ev.record(self.copy)
        sl["window"], sl["ready"], sl["res"] = w, ev, res
        self.uploaded_bytes += n * h.n_feat * 4

    def acquire(self, w: int, also: Sequence[torch.cuda.Stream] = ()) -> G.ResidentPages:
        """Window w as a resident set; the CURRENT stream -- and every stream in ``also`` -- waits (on the device) for its upload.
        Every stream that READS the window must be ordered behind the upload: the batch pipeline assembles on a side stream that
        otherwise waits only for its own buffer events, and would gather page tables, CSRs, labels and (fp32 mode) feature rows
        of a window that is still arriving."""
        if self._slot_of(w) is None:
            self.prefetch(w)
        sl = self._slot_of(w)
        for s in self.slots:
            s["in_use"] = s is sl
        torch.cuda.current_stream(self.device).wait_event(sl["ready"])
        for st in also:
            st.wait_event(sl["ready"])
        return sl["res"]

    def release(self, w: int) -> None:
        """Every step on window w has been queued on the current stream: its slot may be overwritten once they are through."""
        sl = self._slot_of(w)
        if sl is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            sl["free"] = ev
            sl["in_use"] = False


def run_windowed(step, pipe, wp: WindowedPages, stream: WindowStream, n_steps: int, n_global=None, loss_scale=None, on_step=None,
                 host_times: Optional[dict] = None):
    """``n_steps`` steps of ``stream`` (an epoch's worth): per chunk of steps on one window -- acquire it, start the upload of
    the next window of the stream, run the same loop as the all-resident path (models/loop.run_steps).  Returns (last out3,
    nodes of the last step)."""
    from .loop import run_steps
    out3, done, last_nodes = None, 0, 0
    import time
    tick = time.perf_counter
    t_ = tick()
    chunks = stream.take(n_steps)
    if host_times is not None:
        host_times["plan"] = host_times.get("plan", 0.0) + tick() - t_
    for i, (w, steps) in enumerate(chunks):
        t_ = tick()
        res = wp.acquire(w, also=(pipe.side,))       # the assembly stream reads the window too (page tables, CSRs, labels, rows)
        t_a = tick()
        if pipe.res is not res:
            pipe.rebind(res)
        t_b = tick()
        if host_times is not None:
            host_times["acquire"] = host_times.get("acquire", 0.0) + t_a - t_
            host_times["rebind"] = host_times.get("rebind", 0.0) + t_b - t_a
        # the window the stream needs after this one: the next chunk's, or -- at the end of this call -- where the stream stands
        if i + 1 < len(chunks):
            nxt = chunks[i + 1][0]
        else:
            nxt = stream.peek_window() if stream.peek_window() != w else stream.next_window()
        if nxt != w:
            wp.prefetch(nxt)
        if host_times is not None:
            host_times["window"] = host_times.get("window", 0.0) + tick() - t_
            host_times["chunks"] = host_times.get("chunks", 0) + 1
        t_ = tick()
        k = len(steps)
        out3 = run_steps(step, pipe, steps, n_global=None if n_global is None else n_global[done:done + k],
                         loss_scale=None if loss_scale is None else loss_scale[done:done + k], on_step=on_step)
        last_nodes = pipe.nodes(k - 1)
        wp.release(w)
        if host_times is not None:
            host_times["steps"] = host_times.get("steps", 0.0) + tick() - t_
        done += k
    return out3, last_nodes


class OwnedResident:
    """The interface of WindowedPages for a rank whose OWN pages fit its budget: one window = all of them, resident for good.
    (Data-parallel runs under a budget: a rank holds the pages it owns -- set / world bytes --, not the whole set.)"""

    def __init__(self, graphs: Sequence[G.PageGraph], device):
        self.device = torch.device(device)
        self.res = G.ResidentPages(graphs, self.device)
        self.ranges = [(0, len(graphs))]
        self.uploaded_bytes = 0
        nodes = np.array([g.num_nodes() for g in graphs], dtype=np.int64)
        edges = np.array([g.num_edges() for g in graphs], dtype=np.int64)
        self.device_bytes = int(nodes.sum() * WindowedPages.bytes_per_node(nodes, edges, int(graphs[0].ndata['feat'].shape[1]), False))

    def prefetch(self, w: int) -> None:
        pass

    def acquire(self, w: int, also: Sequence[torch.cuda.Stream] = ()) -> G.ResidentPages:
        return self.res

    def to_images(self, agg: bool) -> None:
        """layer 0 takes images: make them now and let the fp32 rows go (graph.ResidentPages.drop_f32)"""
        self.res.enable_p3(agg=agg)
        self.res.drop_f32()

    def release(self, w: int) -> None:
        pass


def default_budget_bytes(set_bytes: float, total_bytes: float) -> Optional[float]:
    """The HBM budget of the training pages when GTE_RESIDENT_BUDGET_GB is not set: None (keep the whole set resident) while the
    set fits in half of the device's memory -- while the images are made the fp32 rows are still there (a third more than the
    resident form), and the rest is for the validation graph, the step's per-batch buffers (a few GB at hidden 1000) and the
    allocator's slack --, otherwise 45 % of it (two window slots + staging rows).  A function of the device's TOTAL memory: what
    other processes hold at launch must not change the tier, the windows and with them the order the pages are visited in."""
    if set_bytes <= 0.5 * total_bytes:
        return None
    return 0.45 * total_bytes
