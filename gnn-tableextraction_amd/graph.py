"""Page-graph container: the DGL-graph duck type the reference's model and train loop use.

The reference builds one ``dgl.graph((u, v), num_nodes, idtype=torch.int32)`` per PDF page
(src/components/graphs/builder.py:425), batches pages with ``dgl.batch``
(src/models/model_train.py:246,297) and the model touches only
``g.local_var() / g.in_degrees() / g.ndata / g.edata / g.update_all(msg, reduce) / g.to(dev)``
(src/components/graphs/models.py:46-78, :105-106, :144-150).  ``PageGraph`` implements that
surface over device-resident CSR arrays so ``model(g)`` runs unchanged without DGL.

Layout in HBM (all int32, the reference's idtype):
  in-edge CSR   ``indptr[N+1], indices[E]`` (= sources, rows = destinations), ``perm[E]`` = COO
                edge ids in row order (stable by destination => fixed summation order)
  out-edge CSR  the same keyed by source (rows = sources, indices = destinations): the backward
                of the aggregation is an SpMM over it
Edge data (``edata['feat']``, one fp32 scalar per edge: loader.py:332-344) stays in COO order as
the user set it; the CSR-ordered copies are cached per tensor version.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

ROW_MAP_PAD = 33       # entries past a batch's row map that name the row past the resident image (gte_gemm_p3_tn_rows reads
                       # the map up to k rounded up to 16, plus 1)


class _FrameDict(dict):
    """ndata / edata: plain dict with DGL's pop semantics."""


class CSR:
    __slots__ = ("indptr", "indices", "perm", "max_deg")

    def __init__(self, indptr, indices, perm):
        self.indptr, self.indices, self.perm = indptr, indices, perm
        self.max_deg = None           # largest row, read from the device on first use (_max_degree)


def _build_csr(key: torch.Tensor, other: torch.Tensor, n: int) -> CSR:
    """Stable counting sort of the COO by ``key`` -> CSR(indptr, indices=other[perm], perm)."""
    from . import ops
    if key.is_cuda:
        return CSR(*ops.coo_to_csr(key, other, n))
    # host-side graph preparation (CPU tensors): index bookkeeping only, no feature arithmetic
    e = key.numel()
    if e == 0:
        z = torch.zeros(n + 1, dtype=torch.int32, device=key.device)
        return CSR(z, key.new_zeros(0, dtype=torch.int32), key.new_zeros(0, dtype=torch.int32))
    perm = torch.sort(key.long(), stable=True).indices
    counts = torch.bincount(key.long(), minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    indptr[1:] = torch.cumsum(counts, 0)
    return CSR(indptr.to(torch.int32), other[perm].to(torch.int32).contiguous(), perm.to(torch.int32))


def _max_degree(csr) -> int:
    if getattr(csr, "max_deg", None) is None:
        ip = csr.indptr
        csr.max_deg = int((ip[1:] - ip[:-1]).max()) if ip.numel() > 1 else 0
    return csr.max_deg


class PageGraph:
    """Directed multigraph over ``num_nodes`` nodes with edges ``src[e] -> dst[e]``."""

    def __init__(self, src, dst, num_nodes: int, device=None):
        src = torch.as_tensor(src)
        dst = torch.as_tensor(dst)
        if src.shape != dst.shape or src.dim() != 1:
            raise ValueError("src and dst must be 1-D tensors of equal length")
        if device is not None:
            src, dst = src.to(device), dst.to(device)
        self._src = src.to(torch.int32).contiguous()
        self._dst = dst.to(torch.int32).contiguous()
        self._n = int(num_nodes)
        self.ndata: Dict[str, torch.Tensor] = _FrameDict()
        self.edata: Dict[str, torch.Tensor] = _FrameDict()
        self._in_csr: Optional[CSR] = None
        self._out_csr: Optional[CSR] = None
        self._inv_deg: Optional[torch.Tensor] = None
        self._wcache: Dict[tuple, torch.Tensor] = {}
        self._in_tiles = None
        self._out_tiles = None
        # LDS-staged aggregation needs per-tile distinct-source lists (one sort/unique per graph): worth it
        # for graphs that are aggregated many times (resident batches, the validation graph, inference)
        self.use_tiles = True
        self.batch_num_nodes_: List[int] = [self._n]
        self.batch_num_edges_: List[int] = [int(self._src.numel())]

    # ---- DGL surface used by the reference -------------------------------------------------
    @property
    def device(self):
        return self._src.device

    def num_nodes(self) -> int:
        return self._n

    number_of_nodes = num_nodes

    def num_edges(self) -> int:
        return int(self._src.numel())

    number_of_edges = num_edges

    def edges(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._src, self._dst

    def batch_num_nodes(self) -> torch.Tensor:
        return torch.tensor(self.batch_num_nodes_, dtype=torch.int64)

    def batch_num_edges(self) -> torch.Tensor:
        return torch.tensor(self.batch_num_edges_, dtype=torch.int64)

    def in_degrees(self) -> torch.Tensor:
        ip = self.in_csr().indptr
        return (ip[1:] - ip[:-1]).to(torch.int64)

    def out_degrees(self) -> torch.Tensor:
        ip = self.out_csr().indptr
        return (ip[1:] - ip[:-1]).to(torch.int64)

    def max_in_degree(self) -> int:
        """Largest in-degree (cached with the CSR, which local_var() copies share: one device->host read per graph): hub rows
        send the aggregation to the edge-parallel kernel (ops.spmm_csr)."""
        return _max_degree(self.in_csr())

    def max_out_degree(self) -> int:
        return _max_degree(self.out_csr())

    def local_var(self) -> "PageGraph":
        """Shallow copy: new ndata/edata dicts over the same tensors and the same cached CSRs
        (models.py:47)."""
        g = PageGraph.__new__(PageGraph)
        g.__dict__.update(self.__dict__)
        g.ndata = _FrameDict(self.ndata)
        g.edata = _FrameDict(self.edata)
        return g

    class _Scope:
        def __init__(self, g):
            self.g = g

        def __enter__(self):
            self.nd, self.ed = _FrameDict(self.g.ndata), _FrameDict(self.g.edata)
            return self.g

        def __exit__(self, *exc):
            self.g.ndata, self.g.edata = self.nd, self.ed
            return False

    def local_scope(self):
        return PageGraph._Scope(self)

    def to(self, device) -> "PageGraph":
        device = torch.device(device)
        if device == self.device:
            return self
        g = PageGraph.__new__(PageGraph)
        g.__dict__.update(self.__dict__)
        g._src, g._dst = self._src.to(device), self._dst.to(device)
        g.ndata = _FrameDict({k: v.to(device) for k, v in self.ndata.items()})
        g.edata = _FrameDict({k: v.to(device) for k, v in self.edata.items()})
        mv = lambda c: None if c is None else CSR(c.indptr.to(device), c.indices.to(device), c.perm.to(device))
        g._in_csr, g._out_csr = mv(self._in_csr), mv(self._out_csr)
        g._inv_deg = None if self._inv_deg is None else self._inv_deg.to(device)
        g._wcache = {}
        g._in_tiles = g._out_tiles = None
        return g

    def update_all(self, message_func, reduce_func) -> None:
        """``update_all(fn.u_mul_e(u, e, m), fn.sum|mean(m, out))`` / ``fn.copy_u`` through the HIP
        aggregation kernel (differentiable w.r.t. the node feature)."""
        from . import ops
        kind = message_func[0]
        if kind not in ("u_mul_e", "copy_u") or reduce_func[0] not in ("sum", "mean"):
            raise NotImplementedError(f"update_all({message_func}, {reduce_func}) is not part of the hot path")
        h = self.ndata[message_func[1]]
        w = self.edata[message_func[2]] if kind == "u_mul_e" else None
        self.ndata[reduce_func[2]] = ops.aggregate(self, h, w, mean=(reduce_func[0] == "mean"))

    # ---- CSR views ---------------------------------------------------------------------------
    def in_csr(self) -> CSR:
        if self._in_csr is None:
            self._in_csr = _build_csr(self._dst, self._src, self._n)
        return self._in_csr

    def out_csr(self) -> CSR:
        if self._out_csr is None:
            self._out_csr = _build_csr(self._src, self._dst, self._n)
        return self._out_csr

    TILES_MIN_NODES = 100_000      # tile plans only pay for graphs whose features overflow the caches (ops.use_tiled)

    def _tiles(self, which: str):
        if not self.use_tiles or not self._src.is_cuda or self._n < self.TILES_MIN_NODES:
            return None
        attr = "_in_tiles" if which == "in" else "_out_tiles"
        plan = getattr(self, attr)
        if plan is None:
            from . import ops
            csr = self.in_csr() if which == "in" else self.out_csr()
            plan = ops.build_tile_plan(csr.indptr, csr.indices, self._n)
            setattr(self, attr, plan)
        return plan

    def in_tiles(self):
        """Tile plan of the in-edge CSR (forward aggregation) or None when tiling is off / not worthwhile."""
        return self._tiles("in")

    def out_tiles(self):
        return self._tiles("out")

    def out_to_in_pos(self) -> torch.Tensor:
        """pos[i] = position in the in-edge CSR of the edge stored at position i of the out-edge CSR (int32 [E])."""
        hit = self._wcache.get(("o2i",))
        if hit is None:
            pin, pout = self.in_csr().perm, self.out_csr().perm
            if pin is None or pout is None:
                raise NotImplementedError("edge permutation is not kept for this graph")
            inv = torch.empty_like(pin)
            inv[pin.long()] = torch.arange(pin.numel(), dtype=pin.dtype, device=pin.device)
            hit = self._wcache[("o2i",)] = inv[pout.long()].contiguous()
        return hit

    def inv_in_degree(self) -> torch.Tensor:
        """norm of models.py:74-78 as a vector: 1/in_degree, 0 where the degree is 0 (fp32 [N])."""
        if self._inv_deg is None:
            deg = self.in_degrees().to(torch.float32)
            inv = torch.where(deg > 0, 1.0 / deg.clamp(min=1), torch.zeros_like(deg))
            self._inv_deg = inv.contiguous()
        return self._inv_deg

    def _cached(self, tag: str, w: torch.Tensor, make):
        key = (tag, w.data_ptr(), w._version, tuple(w.shape))
        hit = self._wcache.get(key)
        if hit is None:
            if len(self._wcache) > 16:
                self._wcache.clear()
            hit = self._wcache[key] = make()
        return hit

    def in_weights(self, w: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """Edge weights in in-edge-CSR order."""
        if w is None:
            return None
        if w.numel() != self.num_edges():
            raise ValueError(f"edge weight has {w.numel()} entries, graph has {self.num_edges()} edges")
        w = w.reshape(-1)
        return self._cached("in", w, lambda: w.detach().to(torch.float32)[self.in_csr().perm.long()].contiguous())

    def out_weights(self, w: Optional[torch.Tensor], mean: bool) -> Optional[torch.Tensor]:
        """Edge weights in out-edge-CSR order; with ``mean`` each is pre-multiplied by
        1/in_degree(dst) so the backward of (sum * norm) is one plain SpMM."""
        csr = self.out_csr()
        if w is None and not mean:
            return None

        def make():
            base = (torch.ones(self.num_edges(), dtype=torch.float32, device=self.device) if w is None
                    else w.detach().reshape(-1).to(torch.float32))[csr.perm.long()]
            if mean:
                base = base * self.inv_in_degree()[csr.indices.long()]
            return base.contiguous()

        if w is None:
            key = ("out_unit_mean",)
            hit = self._wcache.get(key)
            if hit is None:
                hit = self._wcache[key] = make()
            return hit
        return self._cached("out_mean" if mean else "out", w.reshape(-1), make)


# ---- constructors mirroring dgl.graph / dgl.batch ------------------------------------------------
def graph(data, num_nodes: Optional[int] = None, idtype=torch.int32, device=None) -> PageGraph:
    """``dgl.graph((u, v), num_nodes=..., idtype=torch.int32)`` (builder.py:425)."""
    u, v = data
    u, v = torch.as_tensor(u), torch.as_tensor(v)
    if num_nodes is None:
        num_nodes = int(max(u.max().item(), v.max().item())) + 1 if u.numel() else 0
    return PageGraph(u, v, num_nodes, device=device)


def from_edge_index(edge_index: torch.Tensor, num_nodes: int, edge_weight: Optional[torch.Tensor] = None) -> PageGraph:
    """Tensor-level form named by BASELINE.json: ``edge_index[2, E]`` rows = (src; dst)."""
    g = PageGraph(edge_index[0], edge_index[1], num_nodes)
    if edge_weight is not None:
        g.edata["feat"] = edge_weight
    return g


def batch(graphs: Sequence[PageGraph]) -> PageGraph:
    """Block-diagonal union of page graphs (``dgl.batch``: model_train.py:246,297).  Node and edge
    data present in every graph are concatenated.  Cached CSRs of the parts are concatenated with
    offsets instead of re-sorting (rows of a block-diagonal union are the parts' rows)."""
    graphs = list(graphs)
    if not graphs:
        raise ValueError("batch() needs at least one graph")
    dev = graphs[0].device
    n_off, e_off = [0], [0]
    for g in graphs:
        n_off.append(n_off[-1] + g.num_nodes())
        e_off.append(e_off[-1] + g.num_edges())
    src = torch.cat([g._src + n_off[i] for i, g in enumerate(graphs)])
    dst = torch.cat([g._dst + n_off[i] for i, g in enumerate(graphs)])
    out = PageGraph(src, dst, n_off[-1], device=dev)
    out.batch_num_nodes_ = [n for g in graphs for n in g.batch_num_nodes_]
    out.batch_num_edges_ = [n for g in graphs for n in g.batch_num_edges_]
    for frame, name in ((lambda g: g.ndata, "ndata"), (lambda g: g.edata, "edata")):
        keys = set(frame(graphs[0]).keys())
        for g in graphs[1:]:
            keys &= set(frame(g).keys())
        for k in keys:
            getattr(out, name)[k] = torch.cat([frame(g)[k] for g in graphs], dim=0)

    def cat_csr(get):
        parts = [get(g) for g in graphs]
        if any(p is None for p in parts):
            return None
        indptr = torch.cat([parts[0].indptr[:1]] + [p.indptr[1:] + e_off[i] for i, p in enumerate(parts)])
        indices = torch.cat([p.indices + n_off[i] for i, p in enumerate(parts)])
        perm = torch.cat([p.perm + e_off[i] for i, p in enumerate(parts)])
        return CSR(indptr.to(torch.int32), indices.to(torch.int32), perm.to(torch.int32))

    out._in_csr = cat_csr(lambda g: g._in_csr)
    out._out_csr = cat_csr(lambda g: g._out_csr)
    return out


def unbatch_sizes(g: PageGraph) -> List[int]:
    return list(g.batch_num_nodes_)


def upload_rows(parts: Sequence[torch.Tensor], device, out: Optional[torch.Tensor] = None, chunk_bytes: int = 64 << 20,
                threads: int = 8) -> torch.Tensor:
    """Concatenation of host matrices [n_i, F] as ONE fp32 matrix -- on ``device``, or into ``out`` (e.g. a pinned host matrix) --
    without a host-side torch.cat: the parts are copied into two pinned staging buffers in turn (a few host threads: a row copy
    releases the interpreter lock), each uploaded asynchronously while the other is filled."""
    from concurrent.futures import ThreadPoolExecutor
    n = int(sum(int(p.shape[0]) for p in parts))
    f = int(parts[0].shape[1]) if parts else 0
    if out is None:
        out = torch.empty((n, f), dtype=torch.float32, device=device)
    if n == 0 or f == 0:
        return out
    # jobs: (part index, first row of the part, rows, destination row) cut at the staging-buffer boundaries
    rows = max(1, chunk_bytes // (4 * f)) if out.is_cuda else n
    chunks, cur, fill, r0 = [], [], 0, 0
    for i, p in enumerate(parts):
        a, m = 0, int(p.shape[0])
        while a < m:
            take = min(m - a, rows - fill)
            cur.append((i, a, take, fill))
            fill += take
            a += take
            if fill == rows:
                chunks.append((r0, fill, cur))
                r0, cur, fill = r0 + fill, [], 0
    if fill:
        chunks.append((r0, fill, cur))
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        if not out.is_cuda:                                       # host destination: row copies straight into it
            for r0, _, jobs in chunks:
                list(pool.map(lambda j: out[r0 + j[3]:r0 + j[3] + j[2]].copy_(parts[j[0]][j[1]:j[1] + j[2]]), jobs))
            return out
        stage = [torch.empty((rows, f), dtype=torch.float32).pin_memory() for _ in range(min(2, len(chunks)))]
        done = [None, None]
        for c, (r0, cnt, jobs) in enumerate(chunks):
            k = c & 1
            if done[k] is not None:
                done[k].synchronize()                             # (the buffer about to be refilled has left the host)
            buf = stage[k]
            list(pool.map(lambda j: buf[j[3]:j[3] + j[2]].copy_(parts[j[0]][j[1]:j[1] + j[2]]), jobs))
            out[r0:r0 + cnt].copy_(buf[:cnt], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(out.device))      # (the stream the copy runs on: out's device, not the current one)
            done[k] = ev
    torch.cuda.current_stream(out.device).synchronize()
    return out


# ================================================================================================
# Resident pages: the whole dataset concatenated once in HBM; a batch is four kernel launches
# ================================================================================================
class ResidentBatch(PageGraph):
    """A batched graph produced by :meth:`ResidentPages.batch`: the CSRs and the CSR-ordered edge weights are
    gathered on the device from the resident dataset instead of being rebuilt (no COO, no sort)."""

    _empty = {}

    def __init__(self, n, n_edges, in_csr, out_csr, w_in, w_out_mean, device):
        z = ResidentBatch._empty.get(device)
        if z is None:
            z = ResidentBatch._empty[device] = torch.zeros(0, dtype=torch.int32, device=device)
        super().__init__(z, z, n)
        self._in_csr, self._out_csr = in_csr, out_csr
        self._w_in, self._w_out_mean = w_in, w_out_mean
        self._n_edges = int(n_edges)
        self.batch_num_edges_ = [self._n_edges]

    def num_edges(self) -> int:
        return self._n_edges

    number_of_edges = num_edges

    def edges(self):
        raise NotImplementedError("ResidentBatch keeps CSRs only; COO edge lists are not materialised")

    def in_weights(self, w):
        return None if w is None else self._w_in

    def out_weights(self, w, mean: bool):
        if mean:
            return self._w_out_mean
        raise NotImplementedError("ResidentBatch stores the mean-normalised out-edge weights only")


class ResidentPages:
    """All pages of a dataset, concatenated once on the device.

    Replaces the reference's per-step ``dgl.batch(train_batch).to(device)`` (model_train.py:297): features,
    labels, both CSRs and the CSR-ordered edge weights live in HBM for the whole run; ``batch(page_ids)``
    writes the block-diagonal union of the chosen pages with ``gte_batch_csr`` / ``gte_batch_rows``.
    288 GB of HBM hold ~80 M nodes of 831 fp32 features.
    """

    def __init__(self, graphs: Sequence[PageGraph], device, feat_key: str = "feat", label_key: str = "label",
                 weight_key: str = "feat", with_feat: bool = True):
        from . import ops
        self.device = torch.device(device)
        sizes = [g.num_nodes() for g in graphs]
        self.node_off_host = torch.zeros(len(graphs) + 1, dtype=torch.int64)
        self.node_off_host[1:] = torch.cumsum(torch.tensor(sizes), 0)
        cpu_graphs = [g.to("cpu") if g.device.type != "cpu" else g for g in graphs]
        # The feature rows (3 kB per node at F0 = 831) do not go through the host-side concatenation of batch(): that was 8.8 of
        # the 10 s this constructor took for 6 000 pages (torch.cat of 4.7 GB into fresh pageable memory, then a pageable
        # upload).  They are uploaded page by page through two pinned staging buffers (upload_rows); batch() sees the rest.
        light = []
        for g in cpu_graphs:
            h = g.local_var()
            h.ndata.pop(feat_key, None)
            light.append(h)
        whole = batch(light).to(self.device)
        n = whole.num_nodes()
        self.n_pages, self.n_nodes = len(graphs), n
        feats = [g.ndata[feat_key] for g in cpu_graphs]
        self.feat = (upload_rows(feats, self.device) if with_feat
                     else torch.empty((0, int(feats[0].shape[1])), dtype=torch.float32, device=self.device))
        self.feat_p3, self.agg_p3, self.p3_mode = None, None, False
        lab = whole.ndata.get(label_key)
        self.label = None if lab is None else lab.to(torch.float32).reshape(-1, 1).contiguous()
        w = whole.edata.get(weight_key)
        self.weighted = w is not None
        self.node_off = self.node_off_host.to(torch.int32).to(self.device)
        page_of_node = torch.repeat_interleave(torch.arange(len(graphs), device=self.device),
                                               torch.tensor(sizes, device=self.device))
        self.max_deg = {"in": _max_degree(whole.in_csr()), "out": _max_degree(whole.out_csr())}     # (bounds for every batch)
        self._sets = {}
        for name, csr, wt in (("in", whole.in_csr(), whole.in_weights(w)),
                              ("out", whole.out_csr(), whole.out_weights(w, True))):
            edge_off = csr.indptr.long()[self.node_off.long()]                              # [P+1] CSR entries per page
            row_of_entry = torch.repeat_interleave(torch.arange(n, device=self.device),
                                                   (csr.indptr[1:] - csr.indptr[:-1]).long())
            page_of_entry = page_of_node[row_of_entry]
            indices_loc = (csr.indices.long() - self.node_off.long()[page_of_entry]).to(torch.int32).contiguous()
            # packed local indptr: page p's n_p + 1 entries at node_off[p] + p
            pos_page = torch.repeat_interleave(torch.arange(len(graphs), device=self.device),
                                               torch.tensor([s + 1 for s in sizes], device=self.device))
            pos_row = torch.arange(n + len(graphs), device=self.device) - pos_page          # global row (or page end)
            indptr_loc = (csr.indptr.long()[pos_row] - edge_off[pos_page]).to(torch.int32).contiguous()
            self._sets[name] = dict(edge_off=edge_off.to(torch.int32).contiguous(), edge_off_host=edge_off.cpu(),
                                    indices_loc=indices_loc, indptr_loc=indptr_loc,
                                    weight=None if wt is None else wt.contiguous())

    @classmethod
    def from_arrays(cls, device, node_off_host: torch.Tensor, feat: torch.Tensor, label: Optional[torch.Tensor], sets: dict,
                    weighted: bool, max_deg: dict, feat_p3=None, p3_mode=False, node_off_dev=None, agg_p3=None) -> "ResidentPages":
        """A resident set over arrays that already live on the device in this class's layout (models/residency.py: a WINDOW of a
        host-resident dataset, uploaded slice by slice).  ``sets[name]`` = {edge_off (int32, device), edge_off_host (int64, cpu),
        indptr_loc, indices_loc, weight | None}; ``feat`` may be an empty [0, F] placeholder when ``feat_p3`` carries the rows."""
        self = cls.__new__(cls)
        self.device = torch.device(device)
        self.node_off_host = node_off_host
        self.n_pages, self.n_nodes = int(node_off_host.numel() - 1), int(node_off_host[-1])
        self.feat, self.feat_p3, self.agg_p3, self.p3_mode = feat, feat_p3, agg_p3, p3_mode
        self.label = label
        self.weighted = weighted
        # (node_off_dev: already uploaded by the caller -- a pageable host->device copy here would block the host until the
        # stream it is queued on has drained)
        self.node_off = node_off_dev if node_off_dev is not None else node_off_host.to(torch.int32).to(self.device)
        self.max_deg = dict(max_deg)
        self._sets = sets
        return self

    def whole_in_csr(self):
        """(indptr [N + 1], indices [E], weights | None) of the in-edge CSR over ALL resident pages in global row ids, put together
        from the per-page local arrays (temporaries of E entries; used once, by the cached input aggregate)."""
        dev, s = self.device, self._sets["in"]
        P, N = self.n_pages, self.n_nodes
        sizes = (self.node_off[1:] - self.node_off[:-1]).long()
        E = int(s["edge_off_host"][P])
        # (output_size: without it repeat_interleave reads the total back from the device -- a host synchronisation on the copy
        # stream of a window upload, models/residency.py)
        page_of_node = torch.repeat_interleave(torch.arange(P, device=dev), sizes, output_size=N)
        eoff = s["edge_off"].long()
        rows = torch.arange(N, device=dev)
        indptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        indptr[:N] = (s["indptr_loc"].long()[rows + page_of_node] + eoff[page_of_node]).to(torch.int32)
        indptr[N] = eoff[P].to(torch.int32)
        page_of_entry = torch.repeat_interleave(torch.arange(P, device=dev), (eoff[1:] - eoff[:-1]), output_size=E)
        indices = (s["indices_loc"][:E].long() + self.node_off.long()[page_of_entry]).to(torch.int32)
        return indptr, indices, (None if s["weight"] is None else s["weight"][:E])

    def build_agg_image(self, x_f32: Optional[torch.Tensor] = None, out=None):
        """The P3 image of norm . A_w x -- the mean aggregate of the input features over the in-edges (models.py:53-57 for layer 0:
        `g.update_all(u_mul_e, sum)`, `ah * norm`) -- for every resident node.  It is page-local and does not change over a run,
        so it is made ONCE here instead of once per step: the input layer then multiplies [x | ahn] straight from the two resident
        images (gte_gemm_p3_nt_rows2) and its weight gradient needs no transpose aggregation (gte_gemm_p3_tn_rows2)."""
        from . import ops
        x = self.feat if x_f32 is None else x_f32
        if x.shape[0] != self.n_nodes:
            raise ValueError("build_agg_image needs the fp32 feature rows of every resident node")
        indptr, indices, w = self.whole_in_csr()
        if out is None:
            out = ops.P3.empty(self.n_nodes, int(x.shape[1]), self.device, rows_cap=self.n_nodes + 1)
            out.data[self.n_nodes:].zero_()                       # (the zero row the row maps' padding entries name)
        ops.spmm_csr_p3(indptr, indices, w, x, self.n_nodes, mean=True, out=out)
        self.agg_p3 = out
        return out

    def enable_p3(self, agg: bool = False) -> None:
        """Keep the features as a P3 image (three bf16 planes per value, csrc/p3.h: the operand format of the planes GEMMs)
        and assemble batches of image rows from now on: a batch then carries ``feat_p3`` instead of ``ndata['feat']``.
        Called by the train loop when layer 0 of the step engine takes its input as an image.  ``agg``: also keep the image of the
        input's mean aggregate (build_agg_image); batches then carry ``agg_p3`` behind the same row map."""
        if agg and self.agg_p3 is None and self.feat.shape[0] == self.n_nodes:
            self.build_agg_image()
        if self.feat_p3 is None:
            from . import ops
            # one row more than the set has nodes, zero: the row maps' padding entries name it (an image of 4 GB or more is
            # read without a range check, gte_gemm_p3_tn_rows)
            img = ops.P3.empty(self.n_nodes, int(self.feat.shape[1]), self.device, rows_cap=self.n_nodes + 1)
            img.data[self.n_nodes:].zero_()
            self.feat_p3 = ops.p3_from_f32(self.feat, out=img)
        # "rows": a batch names its rows of the RESIDENT image through a row map (no copy of the image rows at all; the input
        # layer's two GEMMs read the image through the map -- 32-bit buffer offsets below 4 GB of image, 64-bit addresses above);
        # "copy": the batch holds a copy of its image rows.  An image handed in from outside (from_arrays) may lack the zero row
        # behind its last node that the 64-bit path needs: it takes the map below 4 GB only.
        rows_ok = (self.feat_p3.data.shape[0] > self.n_nodes or
                   self.feat_p3.data.numel() + self.feat_p3.ldp < (1 << 32) - 4096)
        want = os.environ.get("GTE_P3_ROWS", "1").lower() not in ("0", "off", "false")
        self.p3_mode = "rows" if (rows_ok and want) else "copy"

    def drop_f32(self) -> None:
        """Free the fp32 feature rows once the batches are row maps into the image(s): nothing in the step reads them any more
        (4 of the 16 bytes a resident feature value costs with both images).  ``disable_p3`` brings them back from the image,
        which holds exactly the fp32 values."""
        if self.feat_p3 is not None and self.p3_mode == "rows" and self.feat.shape[0] == self.n_nodes:
            self.feat = torch.empty((0, int(self.feat.shape[1])), dtype=torch.float32, device=self.device)

    def disable_p3(self) -> None:
        """batches carry fp32 ``ndata['feat']`` again (the image stays cached for the next switch)"""
        if self.feat.shape[0] != self.n_nodes and self.feat_p3 is not None:
            from . import ops
            self.feat = ops.p3_to_f32(ops.P3(self.feat_p3.data[:self.n_nodes], self.n_nodes, int(self.feat.shape[1])))
        self.p3_mode = False

    def __len__(self):
        return self.n_pages

    def page_sizes(self):
        return (self.node_off_host[1:] - self.node_off_host[:-1]).tolist()

    # ---- batch assembly ------------------------------------------------------------------------
    def batch_meta(self, page_ids):
        """Host-side metadata of one batch: (int32 vector [ids | node offsets | in-edge offsets | out-edge offsets],
        n_nodes, n_in_entries, n_out_entries, per-page node counts).  Pure index arithmetic on the page table."""
        ids = torch.as_tensor(page_ids, dtype=torch.int64)
        nb = ids.numel()
        n_sizes = self.node_off_host[ids + 1] - self.node_off_host[ids]
        offs = [torch.zeros(nb + 1, dtype=torch.int64)]
        offs[0][1:] = torch.cumsum(n_sizes, 0)
        for name in ("in", "out"):
            eo = self._sets[name]["edge_off_host"]
            o = torch.zeros(nb + 1, dtype=torch.int64)
            o[1:] = torch.cumsum(eo[ids + 1] - eo[ids], 0)
            offs.append(o)
        meta = torch.cat([ids] + offs).to(torch.int32)
        return meta, int(offs[0][-1]), int(offs[1][-1]), int(offs[2][-1]), n_sizes

    def alloc_batch_buffers(self, cap_nodes: int, cap_in: int, cap_out: int) -> dict:
        """Output buffers of :meth:`assemble` for batches of up to the given sizes (views of them are handed out)."""
        dev = self.device
        i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        return {"cap": (int(cap_nodes), int(cap_in), int(cap_out)),
                "indptr": [i32(cap_nodes + 1), i32(cap_nodes + 1)], "indices": [i32(cap_in), i32(cap_out)],
                "weight": [f32(max(int(cap_in), 1)) if self.weighted else None,
                           f32(max(int(cap_out), 1)) if self._sets["out"]["weight"] is not None else None],
                "feat": None if self.p3_mode else f32(max(int(cap_nodes), 1), self.feat.shape[1]),
                "feat_p3": torch.empty((max(int(cap_nodes), 1), self.feat_p3.ldp), dtype=torch.uint8, device=dev)
                if self.p3_mode == "copy" else None,
                "feat_rows": i32(cap_nodes + ROW_MAP_PAD) if self.p3_mode == "rows" else None,
                "label": None if self.label is None else f32(max(int(cap_nodes), 1), 1)}

    def assemble(self, meta_dev: torch.Tensor, nb: int, n_out: int, e_in: int, e_out: int, bufs: dict, n_sizes=None,
                 stream=None) -> "ResidentBatch":
        """Writes the block-diagonal union described by ``meta_dev`` (device copy of :meth:`batch_meta`'s vector) into
        ``bufs`` with ONE launch of gte_batch_assemble on ``stream`` (default: the current stream) and returns it as a
        graph whose tensors are row views of the buffers.  No allocation, no host synchronisation."""
        lib, P = _lib.load(), _lib.ptr
        st = _lib.current_stream() if stream is None else stream
        cn, ci, co = bufs["cap"]
        if n_out > cn or e_in > ci or e_out > co:
            raise ValueError(f"batch of {n_out} nodes / {e_in}+{e_out} entries exceeds the buffers' capacity {bufs['cap']}")
        pages = meta_dev[:nb]
        b_node, b_ein, b_eout = (meta_dev[nb + i * (nb + 1): nb + (i + 1) * (nb + 1)] for i in range(3))
        csrs, weights, descs = [], [], []
        for k, (name, b_eoff, e_cnt) in enumerate((("in", b_ein, e_in), ("out", b_eout, e_out))):
            s = self._sets[name]
            indptr, indices = bufs["indptr"][k][:n_out + 1], bufs["indices"][k][:e_cnt]
            wout = bufs["weight"][k][:e_cnt] if s["weight"] is not None else None
            descs.append(_lib.BatchArrays(P(s["edge_off"]), P(s["indptr_loc"]), P(s["indices_loc"]), P(s["weight"]) or None,
                                          P(b_eoff), P(indptr), P(indices), P(wout) or None))
            csrs.append(CSR(indptr, indices, None))
            csrs[-1].max_deg = self.max_deg[name]        # (no device read per batch)
            weights.append(wout)
        f = self.feat.shape[1]
        p3 = bufs.get("feat_p3") is not None
        rows = bufs.get("feat_rows")
        lab = None if self.label is None else bufs["label"][:n_out]
        # ONE launch: features, labels, both CSRs and their weights (per-page contiguous runs, 16-byte accesses).  In image
        # mode a feature row is the ldp bytes of its P3 image, moved as ldp / 4 words.
        import ctypes
        if rows is not None:
            fsrc, fld, fcols, fdst = None, 0, 0, None
        elif p3:
            src, feat = self.feat_p3.data, bufs["feat_p3"]
            fsrc, fld, fcols, fdst = P(src), src.stride(0) // 4, src.stride(0) // 4, P(feat)
        else:
            feat = bufs["feat"][:n_out]
            fsrc, fld, fcols, fdst = P(self.feat), self.feat.stride(0), f, P(feat)
        if rows is not None:
            _lib.check(lib.gte_batch_assemble_rows(P(pages), nb, P(self.node_off), P(b_node), ctypes.addressof(descs[0]),
                                                   ctypes.addressof(descs[1]), fsrc, fld, fcols, fdst,
                                                   P(self.label) or None, P(lab) or None, n_out, P(rows), ROW_MAP_PAD, self.n_nodes, st),
                       "gte_batch_assemble_rows")
        else:
            _lib.check(lib.gte_batch_assemble(P(pages), nb, P(self.node_off), P(b_node), ctypes.addressof(descs[0]),
                                              ctypes.addressof(descs[1]), fsrc, fld, fcols, fdst,
                                              P(self.label) or None, P(lab) or None, n_out, st), "gte_batch_assemble")
        g = ResidentBatch(n_out, e_in, csrs[0], csrs[1], weights[0], weights[1], self.device)
        if n_sizes is not None:
            g.batch_num_nodes_ = n_sizes.tolist() if hasattr(n_sizes, "tolist") else list(n_sizes)
        if rows is not None:
            from . import ops
            g.feat_p3 = ops.P3(self.feat_p3.data, n_out, f, row_map=rows, res_rows=self.n_nodes)
            if self.agg_p3 is not None:
                g.agg_p3 = ops.P3(self.agg_p3.data, n_out, f, row_map=rows, res_rows=self.n_nodes)
        elif p3:
            from . import ops
            g.feat_p3 = ops.P3(feat, n_out, f)           # no ndata['feat']: a consumer that needs fp32 rows fails loudly
        else:
            g.ndata["feat"] = feat
        if lab is not None:
            g.ndata["label"] = lab.reshape(-1)
        if self.weighted:
            g.edata["feat"] = weights[0]                # CSR order; ResidentBatch.in/out_weights ignore the argument
        g._keepalive = (meta_dev, bufs)
        return g

    def batch(self, page_ids) -> ResidentBatch:
        """One batch into freshly allocated buffers (tests, one-off use).  The train loop goes through
        ``models.loop.BatchPipeline``: epoch-wide metadata upload, reused buffers, assembly on a side stream."""
        meta, n_out, e_in, e_out, n_sizes = self.batch_meta(page_ids)
        nb = int(torch.as_tensor(page_ids).numel())
        meta_dev = meta.to(self.device, non_blocking=True)                                    # one small H2D copy
        return self.assemble(meta_dev, nb, n_out, e_in, e_out, self.alloc_batch_buffers(n_out, e_in, e_out), n_sizes)


def edge_weights_from_boxes(bbox: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, graph_of_node: torch.Tensor,
                            n_graphs: int) -> torch.Tensor:
    """``edata['feat']`` of loader.py:332-344 on the device: 1 - d/max d per page, d = the reference box distance."""
    from . import ops
    _lib.require_device(bbox, "edge_weights_from_boxes")
    lib = _lib.load()
    bbox = bbox.to(torch.int32).contiguous()
    src, dst = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
    gon = graph_of_node.to(torch.int32).contiguous()
    e = src.numel()
    w = torch.empty(e, dtype=torch.float32, device=bbox.device)
    ws = ops._workspace(lib.gte_edge_weights_workspace_bytes(e, n_graphs), bbox.device, "ew")
    _lib.check(lib.gte_edge_weights_bbox(_lib.ptr(bbox), _lib.ptr(src), _lib.ptr(dst), _lib.ptr(gon), e, n_graphs,
                                         _lib.ptr(w), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "gte_edge_weights_bbox")
    return w


# ================================================================================================
# Page graphs from word boxes, on the device (SURVEY 8(f) N4 + N1 + N3)
# ================================================================================================
def bbox_features(bbox: torch.Tensor, char_counts: torch.Tensor) -> torch.Tensor:
    """The 13 BBOX node features of src/components/nlp/bbox.py:49-124 (9 geometry values + the 4-bin character histogram)
    as float32 [N, 13] -- ``_generate_features(...)`` + ``.float()`` of model_train.py:293-296 for the BBOX embedder.
    ``char_counts`` int32 [N, 3] = (#letters, #digits, #others) of each word without spaces: classifying characters is
    Unicode-table work (str.isalpha / str.isdigit) and stays on the host; every arithmetic step runs in gte_bbox_features."""
    _lib.require_device(bbox, "bbox_features")
    bbox = bbox.to(torch.int32).contiguous()
    counts = char_counts.to(torch.int32).contiguous()
    out = torch.empty((bbox.shape[0], 13), dtype=torch.float32, device=bbox.device)
    _lib.check(_lib.load().gte_bbox_features(_lib.ptr(bbox), _lib.ptr(counts), _lib.ptr(out), 13, bbox.shape[0],
                                             _lib.current_stream()), "gte_bbox_features")
    return out


def knn_graph_from_boxes(bbox: torch.Tensor, node_off, page_size, k: int = 5, max_dist: int = 500, bidirectional: bool = True,
                         labels: Optional[torch.Tensor] = None, range_island: int = 0, text_label: int = 1,
                         edge_features: bool = True, mode: str = "knn"):
    """boxes -> batched page graph (``PREPROCESS.mode`` = 'knn' or 'visibility'), entirely on the device.

    mode='visibility' (builder.py:294-379): every node's nearest visible box to the top / right / bottom / left, vertical edges
    that cross a horizontal edge removed (gte_visibility_select), then the same to_simple + to_bidirected CSR, island removal
    and edge weights as the k-NN mode; ``k`` is ignored, the graph is always bidirected.

    What ``GraphBuilder.get_graph(mode='knn')`` (builder.py:240-292,383-411) and ``Papers2Graphs.modify_graphs``
    (loader.py:296-344: fast_remove_islands, to_simple + to_bidirected, edge weights) do page by page in Python, for all pages
    at once: gte_knn_select -> gte_knn_csr -> gte_island_mask (optional) -> gte_edge_weights_bbox.  ``bbox`` int32 [N, 4] (pages
    concatenated), ``node_off`` [P + 1], ``page_size`` [P, 2] = (width, height) of each page image.
    Returns (graph, keep): a block-diagonal :class:`PageGraph` whose in-edge CSR is already built (sources ascending inside a
    row), ``edata['feat']`` = the edge weights, ``ndata['bbox']`` (and ``ndata['label']``) of the KEPT nodes; ``keep`` = bool [N]
    over the input nodes (all True unless islands were removed).  One host synchronisation (the edge count)."""
    from . import ops
    _lib.require_device(bbox, "knn_graph_from_boxes")
    lib, P = _lib.load(), _lib.ptr
    dev = bbox.device
    bbox = bbox.to(torch.int32).contiguous()
    node_off_h = torch.as_tensor(node_off, dtype=torch.int64).cpu()
    sizes = (node_off_h[1:] - node_off_h[:-1])
    n, n_pages = int(node_off_h[-1]), sizes.numel()
    if bbox.shape[0] != n:
        raise ValueError(f"{bbox.shape[0]} boxes, node_off says {n}")
    node_off_d = node_off_h.to(torch.int32).to(dev)
    page_size_d = torch.as_tensor(page_size, dtype=torch.int32).reshape(-1, 2).to(dev).contiguous()
    page_of_node = torch.repeat_interleave(torch.arange(n_pages, dtype=torch.int32, device=dev), sizes.to(dev))

    def csr_of(sel, bidir):
        indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        ws = ops._workspace(lib.gte_knn_csr_workspace_bytes(n), dev, "knn")
        _lib.check(lib.gte_knn_csr(P(sel), P(node_off_d), P(page_of_node), n, k, int(bidir), 0, P(indptr), None, None, P(ws),
                                   ws.numel(), _lib.current_stream()), "gte_knn_csr")
        e = int(indptr[-1].item())                              # the one synchronisation: sizes of the edge arrays
        indices = torch.empty(max(e, 1), dtype=torch.int32, device=dev)[:e]
        dst_of = torch.empty(max(e, 1), dtype=torch.int32, device=dev)[:e]
        if e:
            _lib.check(lib.gte_knn_csr(P(sel), P(node_off_d), P(page_of_node), n, k, int(bidir), 1, P(indptr), P(indices),
                                       P(dst_of), None, 0, _lib.current_stream()), "gte_knn_csr fill")
        return indptr, indices, dst_of

    if mode == "visibility":
        if not bidirectional:
            raise ValueError("the visibility graph is built bidirected (loader.py:319-320); bidirectional=False is a k-NN option")
        k = 4                                                    # top, right, bottom, left
        sel = torch.empty((n, 4), dtype=torch.int32, device=dev)
        _lib.check(lib.gte_visibility_select(P(bbox), P(node_off_d), P(page_size_d), n_pages, n, int(sizes.max()) if n_pages else 0,
                                             int(max_dist), P(sel), _lib.current_stream()), "gte_visibility_select")
    elif mode == "knn":
        sel = torch.empty((n, k), dtype=torch.int32, device=dev)
        _lib.check(lib.gte_knn_select(P(bbox), P(node_off_d), P(page_size_d), n_pages, n, int(sizes.max()) if n_pages else 0, k,
                                      int(max_dist), P(sel), _lib.current_stream()), "gte_knn_select")
    else:
        raise ValueError("mode should be either 'visibility' or 'knn'")        # builder.py:409
    indptr, indices, dst_of = csr_of(sel, bidirectional)
    keep = torch.ones(n, dtype=torch.bool, device=dev)
    if range_island and labels is not None:
        lab32 = labels.to(dev).to(torch.int32).contiguous()
        # fast_remove_islands asserts that a page has a non-TEXT node (builder.py:576: 'only text in graph'); without the check
        # every node of such a page would be an island and the page would silently vanish
        non_text = torch.zeros(n_pages, dtype=torch.int64, device=dev).index_add_(0, page_of_node.long(), (lab32 != int(text_label)).to(torch.int64))
        empty_ok = torch.as_tensor(sizes).to(dev) == 0
        bad = torch.nonzero((non_text == 0) & ~empty_ok).flatten()
        if bad.numel():
            raise ValueError(f"ERROR - only text in graph -> what to do? (page {int(bad[0])}: island removal needs a non-TEXT node; "
                             f"builder.py:576 asserts the same)")
        sym = (indptr, indices) if bidirectional else csr_of(sel, True)[:2]
        island = torch.empty(n, dtype=torch.uint8, device=dev)
        ws = ops._workspace(2 * n + 256, dev, "island")
        _lib.check(lib.gte_island_mask(P(sym[0]), P(sym[1]), P(lab32), n, int(range_island), int(text_label), P(island), P(ws),
                                       ws.numel(), _lib.current_stream()), "gte_island_mask")
        keep = island == 0
        if not bool(keep.all()):
            # induced subgraph on the kept nodes (g.remove_nodes(to_remove), loader.py:300-301): index bookkeeping on the device
            new_id = (torch.cumsum(keep.to(torch.int32), 0) - 1).to(torch.int32)
            ok = keep[indices.long()] & keep[dst_of.long()]
            indices, dst_of = new_id[indices[ok].long()].contiguous(), new_id[dst_of[ok].long()].contiguous()
            kept_sizes = torch.zeros(n_pages, dtype=torch.int64, device=dev).index_add_(0, page_of_node.long(), keep.to(torch.int64))
            sizes = kept_sizes.cpu()
            bbox = bbox[keep].contiguous()
            page_of_node = page_of_node[keep].contiguous()
            n = int(sizes.sum())
            deg = torch.zeros(n, dtype=torch.int32, device=dev).index_add_(0, dst_of.long(), torch.ones_like(dst_of))
            indptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
            indptr[1:] = torch.cumsum(deg, 0)
    g = PageGraph(indices, dst_of, n, device=dev)
    g._in_csr = CSR(indptr, indices, torch.arange(indices.numel(), dtype=torch.int32, device=dev))   # COO is in row order
    g.batch_num_nodes_ = sizes.tolist()
    g.ndata["bbox"] = bbox
    if labels is not None:
        g.ndata["label"] = labels.to(dev)[keep] if not bool(keep.all()) else labels.to(dev)
    if edge_features:
        g.edata["feat"] = edge_weights_from_boxes(bbox, indices, dst_of, page_of_node, n_pages)
    return g, keep
