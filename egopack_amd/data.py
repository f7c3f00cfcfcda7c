"""Host-side data plumbing of the hot path: sample / batch containers, temporal edge builders,
CSR construction, collation, the multi-task loader and synthetic Omnivore-shaped datasets.

Everything here is integer / index work on the host (the reference does it in PyG dataloader
workers): results are bit-exact with the reference semantics.
No torch_geometric dependency.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch


# --------------------------------------------------------------------------------------------
# containers
# --------------------------------------------------------------------------------------------
class Data:
    """Attribute bag with the fields the reference reads from a PyG ``Data``/``Batch``:
    x [N,S,F], pos [N], y, edge_index [2,E], batch [N], ptr [B+1], num_graphs, plus the cached
    ``graph`` (CSRGraph) the HIP aggregation kernels consume."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __contains__(self, key):
        return getattr(self, key, None) is not None

    def materialise(self):
        """Every field present in ``__dict__`` (a ``LazyData`` builds its tensor views here); what walks ``__dict__`` calls it."""
        return self

    def keys(self):
        return [k for k, v in self.materialise().__dict__.items() if v is not None]

    def to(self, device, non_blocking: bool = False):
        out = Data()
        for k, v in self.materialise().__dict__.items():
            if k == "_arena":
                continue
            if torch.is_tensor(v) or isinstance(v, CSRGraph):
                v = v.to(device, non_blocking=non_blocking)
            elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
                v = [t.to(device, non_blocking=non_blocking) for t in v]
            setattr(out, k, v)
        return out

    def pin_memory(self):
        for k, v in self.materialise().__dict__.items():
            if torch.is_tensor(v) or isinstance(v, CSRGraph):
                setattr(self, k, v.pin_memory())
        return self


class LazyData(Data):
    """A batch of a packed transfer (``to_device_packed``) whose ~20 tensor views are built on FIRST USE: a training loop that
    replays a captured step takes the transfer's byte buffer as a whole (engine.StepBase.train_step) and never looks at the
    individual tensors, and building them costs more host time per step than the copy they describe.  ``fill()`` returns the
    fields; attributes assigned before the first use (the feature block, fingerprints) take precedence."""

    fills = 0  # batches whose views were built (tests: a replayed training step builds none)

    def __init__(self, fill):
        self.__dict__["_fill"] = fill

    def materialise(self):
        fill = self.__dict__.pop("_fill", None)
        if fill is not None:
            LazyData.fills += 1
            for k, v in fill().items():
                self.__dict__.setdefault(k, v)
        return self

    def __getattr__(self, name):  # (reached only for a name that is not in __dict__)
        if name.startswith("__") or "_fill" not in self.__dict__:
            raise AttributeError(name)
        self.materialise()
        try:
            return self.__dict__[name]
        except KeyError:
            raise AttributeError(name) from None


HEAVY_DEGREE = 24  # = egk_csr_heavy_threshold() (tests/test_cabi.py checks the two agree)
HEAVY_IN_LAUNCH_DEGREE = 64  # listed rows up to this many edges: one workgroup each inside the gather launch (heavy_mode 1)


@dataclass
class CSRGraph:
    """Both orientations of an edge list as int32 CSR.

    rowptr/col : in-edges grouped by TARGET (col = source)      -> forward mean aggregation
    t_rowptr/t_col/t_wgt : out-edges grouped by SOURCE (t_col = target, t_wgt = 1/in_degree(target))
                                                               -> its backward, also a gather
    Inside a row, entries keep the edge_index order (stable sort), i.e. the order in which the
    reference's scatter would visit them."""
    rowptr: torch.Tensor
    col: torch.Tensor
    t_rowptr: torch.Tensor
    t_col: torch.Tensor
    t_wgt: torch.Tensor
    num_nodes: int
    # ascending ids of the rows with more than HEAVY_DEGREE edges in each orientation (usually empty; the LTA fan-out
    # node has out-degree T - 1): the gather kernel cuts those rows over several workgroups (egk_csr_gather)
    heavy: Optional[torch.Tensor] = None
    t_heavy: Optional[torch.Tensor] = None
    # egk_csr_gather's heavy_mode per orientation: 1 if every listed row has at most HEAVY_IN_LAUNCH_DEGREE edges (T = 32:
    # the fan-out node's 31), 0 if some have hundreds (T = 256)
    heavy_mode: int = 0
    t_heavy_mode: int = 0
    # uint8 [num_nodes]: the in-neighbours of row i when they are a subset of {i - 1, i, i + 1} in ascending order (bit 0:
    # i - 1, bit 1: i, bit 2: i + 1), 0xFF for any other row -- egk_csr_gather_banded reads the neighbours of a coded row
    # without fetching rowptr / col (a radius-1 temporal graph is banded everywhere but at the LTA forecast nodes)
    band: Optional[torch.Tensor] = None

    def _map(self, f):
        return CSRGraph(*(f(t) for t in (self.rowptr, self.col, self.t_rowptr, self.t_col, self.t_wgt)), self.num_nodes,
                        *(f(t) if t is not None else None for t in (self.heavy, self.t_heavy)), self.heavy_mode, self.t_heavy_mode,
                        f(self.band) if self.band is not None else None)

    def to(self, device, non_blocking: bool = False):
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def pin_memory(self):
        return self._map(lambda t: t.pin_memory())


def build_csr(edge_index: torch.Tensor, num_nodes: int) -> CSRGraph:
    """edge_index [2,E] (row 0 = source j, row 1 = target i; flow source->target)."""
    dev = edge_index.device
    src, tgt = edge_index[0].long(), edge_index[1].long()
    deg_in = torch.bincount(tgt, minlength=num_nodes)
    order = torch.argsort(tgt, stable=True)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg_in, 0)
    col = src[order]
    deg_out = torch.bincount(src, minlength=num_nodes)
    t_order = torch.argsort(src, stable=True)
    t_rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
    t_rowptr[1:] = torch.cumsum(deg_out, 0)
    t_col = tgt[t_order]
    t_wgt = 1.0 / deg_in[t_col].clamp(min=1).to(torch.float32)
    heavy = torch.nonzero(deg_in > HEAVY_DEGREE).flatten().int()
    t_heavy = torch.nonzero(deg_out > HEAVY_DEGREE).flatten().int()
    mode = lambda deg, listed: int(listed.numel() > 0 and int(deg.max()) <= HEAVY_IN_LAUNCH_DEGREE)
    return CSRGraph(rowptr.int(), col.int(), t_rowptr.int(), t_col.int(), t_wgt, int(num_nodes), heavy, t_heavy,
                    mode(deg_in, heavy), mode(deg_out, t_heavy), band_codes(rowptr, col, num_nodes))


def band_codes(rowptr: torch.Tensor, col: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """uint8 [num_nodes] neighbour codes of a by-target CSR (CSRGraph.band): for a row whose entries are, in this order, a
    subset of (i - 1, i, i + 1) the OR of bit 0 / 1 / 2; 0xFF for every other row (more than three entries, an entry farther
    away, or entries out of ascending order)."""
    n = int(num_nodes)
    deg = (rowptr[1:] - rowptr[:-1]).long()
    row_of = torch.repeat_interleave(torch.arange(n, device=col.device), deg)
    off = col.long() - row_of  # -1, 0, +1 for band entries
    near = off.abs() <= 1
    bad = torch.zeros(n, dtype=torch.bool, device=col.device)
    bad |= deg > 3
    if col.numel():
        bad.index_put_((row_of[~near],), torch.ones(int((~near).sum()), dtype=torch.bool, device=col.device), accumulate=False)
        same_row = row_of[1:] == row_of[:-1]
        unordered = same_row & (off[1:] <= off[:-1])  # (also catches duplicates)
        bad.index_put_((row_of[1:][unordered],), torch.ones(int(unordered.sum()), dtype=torch.bool, device=col.device), accumulate=False)
    bits = torch.zeros(n, dtype=torch.int64, device=col.device)
    if col.numel():
        ok = near
        bits.index_add_(0, row_of[ok], (1 << (off[ok] + 1)).long())
    code = torch.where(bad, torch.full_like(bits, 0xFF), bits.clamp(max=7))
    return code.to(torch.uint8)


# --------------------------------------------------------------------------------------------
# edge builders (reference a17)
# --------------------------------------------------------------------------------------------
def radius_band_edges(pos: torch.Tensor, k: int, max_num_neighbors: int = 32) -> torch.Tensor:
    """RadiusGraph(r=k+0.5, loop=False) on the 1-D clip positions of ONE sample (reference
    main_temporal.py:168): all (j -> i), i != j, |pos_i - pos_j| <= k.  Grouped by target, ascending
    source inside a group, at most ``max_num_neighbors`` per target."""
    p = pos.reshape(-1).to(torch.int64)
    n = p.numel()
    d = (p.view(-1, 1) - p.view(1, -1)).abs()
    adj = (d.to(torch.float64) <= (k + 0.5)) & ~torch.eye(n, dtype=torch.bool)
    tgt, src = adj.nonzero(as_tuple=True)
    if max_num_neighbors is not None and n > max_num_neighbors:
        deg = torch.bincount(tgt, minlength=n)
        start = torch.cumsum(deg, 0) - deg
        keep = (torch.arange(tgt.numel()) - start[tgt]) < max_num_neighbors
        tgt, src = tgt[keep], src[keep]
    return torch.stack([src, tgt])


def lta_connectivity_edges(pos: torch.Tensor, y: torch.Tensor, r: float) -> torch.Tensor:
    """LTATemporalConnectivity (reference models/transforms/lta_temp_connectivity.py:30-56): radius
    band of radius r plus (last floor(r) input clips) -> (forecast clips), coalesced (sorted by
    source*N+target, duplicates removed).  n_forecast counts ``y[:,0] > 0`` exactly as the reference
    does (a verb label 0 is not counted)."""
    n = pos.shape[0]
    band = radius_band_edges(pos, int(math.floor(r)))
    n_in = int((y[:, 0] == -1).sum())
    n_f = int((y[:, 0] > 0).sum())
    lo = max(math.ceil(n_in - r), 0)
    src = torch.arange(lo, n_in, dtype=torch.long).repeat_interleave(n_f)
    tgt = torch.arange(n_in, n_in + n_f, dtype=torch.long).repeat(min(math.floor(r), n_in))
    ei = torch.cat([torch.stack([src, tgt]), band], dim=-1)
    key = torch.unique(ei[0] * n + ei[1], sorted=True)
    return torch.stack([key // n, key % n])


class RadiusGraph:
    """Callable transform with the constructor of torch_geometric.transforms.RadiusGraph."""

    def __init__(self, r: float, loop: bool = False, max_num_neighbors: int = 32, flow: str = "source_to_target"):
        if loop or flow != "source_to_target":
            raise ValueError("only loop=False, flow='source_to_target' are on the hot path")
        self.r, self.max_num_neighbors = r, max_num_neighbors

    def __call__(self, data: Data) -> Data:
        data.edge_attr = None
        data.edge_index = radius_band_edges(data.pos, int(math.floor(self.r)), self.max_num_neighbors)
        return data


class LTATemporalConnectivity:
    """Same constructor / call contract as the reference transform."""

    def __init__(self, r: float, loop: bool = False, max_num_neighbors: int = 32, flow: str = "source_to_target",
                 num_workers: int = 1):
        self.r = r

    def __call__(self, data: Data) -> Data:
        if getattr(data, "batch", None) is not None:
            raise ValueError("This transform expects no batched graphs.")
        data.edge_attr = None
        data.edge_index = lta_connectivity_edges(data.pos, data.y, self.r)
        return data


# --------------------------------------------------------------------------------------------
# labelled rows of a per-node multi-head label tensor (the heads' row compaction, engine.MTLStep)
# --------------------------------------------------------------------------------------------
import os as _os
# compact only when at most this share of the nodes carries a label (EGK_LIVE_SHARE: development knob)
LIVE_ROWS_MAX_SHARE = float(_os.environ.get("EGK_LIVE_SHARE", "0.75"))


def live_label_rows(y, n_nodes: int):
    """The reference labels only some nodes of a task batch: AR the centre node of every sequence (data/ego4d_fho.py:222-223),
    LTA everything but the two input nodes (:361-362); every other node carries ``ignore_index`` in every head, so its loss is 0
    and its gradient is 0 (criterion/wrapper.py:67-82, nn.CrossEntropyLoss(ignore_index=-1)) -- while the heads
    (models/tasks/task.py:17-26, recognition.py:39-59) are ROW-WISE.  A training step can therefore run the heads on the
    labelled rows only.  Host-side integer work: for ``y`` [N, heads] int64 returns (live_idx [cap] int64, live_inv [N] int64,
    live_y [cap, heads]) -- the labelled rows in node order, padded with -1 (an all-zero row / an all-ignored label) to a
    multiple of 64 rows -- or None when y is not such a tensor or more than LIVE_ROWS_MAX_SHARE of the nodes are labelled."""
    import numpy as np
    if not torch.is_tensor(y) or y.dim() != 2 or y.shape[0] != n_nodes or n_nodes == 0 or y.is_floating_point() or y.is_cuda:
        return None
    ya = y.numpy()
    rows = np.flatnonzero((ya != -1).any(axis=1))
    if rows.size > LIVE_ROWS_MAX_SHARE * n_nodes:
        return None
    cap = max(64, (rows.size + 63) // 64 * 64)
    idx = np.full(cap, -1, dtype=np.int64)
    idx[:rows.size] = rows
    inv = np.full(n_nodes, -1, dtype=np.int64)
    inv[rows] = np.arange(rows.size, dtype=np.int64)
    yl = np.full((cap, ya.shape[1]), -1, dtype=np.int64)
    yl[:rows.size] = ya[rows]
    return torch.from_numpy(idx), torch.from_numpy(inv), torch.from_numpy(yl)


def live_rows_progression(live_idx):
    """(first, step, count) when the labelled rows are an arithmetic progression that fills the padded list (count a multiple of
    64: one labelled node per sequence of equal length -- AR) -- a strided view then stands for the gathered rows -- else None."""
    idx = live_idx.numpy() if torch.is_tensor(live_idx) else live_idx
    n = int(idx.shape[0])
    if n < 2 or idx[-1] < 0:
        return None
    step = int(idx[1] - idx[0])
    if step < 1 or not bool((idx[1:] - idx[:-1] == step).all()):
        return None
    return int(idx[0]), step, n


def _attach_live_rows(out: "Data") -> None:
    lr = live_label_rows(out.y, int(out.pos.shape[0]))
    if lr is not None:
        out.live_idx, out.live_inv, out.live_y = lr
        ap = live_rows_progression(out.live_idx)
        if ap is not None:
            out.live_ap = ap  # (host integers: known without a device round trip, part of a captured step's signature)


# --------------------------------------------------------------------------------------------
# collation (reference a18: PyG Batch.from_data_list)
# --------------------------------------------------------------------------------------------
def collate(samples: Sequence[Data], with_csr: bool = True) -> Data:
    xs, ys, poss, eis, batch, ptr = [], [], [], [], [], [0]
    off = 0
    resident = getattr(samples[0], "x", None) is None and getattr(samples[0], "x_idx", None) is not None
    for b, s in enumerate(samples):
        n = s.pos.shape[0]
        xs.append(s.x_idx if resident else s.x)
        poss.append(s.pos)
        ys.append(s.y if torch.is_tensor(s.y) else torch.tensor([s.y]))
        eis.append(s.edge_index + off)
        batch.append(torch.full((n,), b, dtype=torch.long))
        off += n
        ptr.append(off)
    ei = torch.cat(eis, dim=1)
    out = Data(x=None if resident else torch.cat(xs), y=torch.cat(ys), pos=torch.cat(poss), edge_index=ei,
               batch=torch.cat(batch), ptr=torch.tensor(ptr, dtype=torch.long), num_graphs=len(samples))
    if out.pos.numel() and not out.pos.is_floating_point():  # host integers: the positional-encoding table's range
        out.pos_range = (int(out.pos.min()), int(out.pos.max()))
    if resident:  # rows of the device-resident feature store (feature_store.FeatureStore.gather builds x in HBM)
        out.x_idx = torch.cat(xs)
    out.graph = build_csr(ei, off) if with_csr else None
    out.ptr32 = out.ptr.to(torch.int32)
    out.seg_ptr = torch.tensor([0, off], dtype=torch.int32)  # one graph-LayerNorm segment: the whole batch
    # per-sample scalar attributes (pnr_frame, start_frame, end_frame, ...) become [B] tensors, as PyG's collation
    # does for python numbers (utils/dataloading.py:56-70; read by the PNR meter)
    _attach_live_rows(out)
    known = {"x", "x_idx", "y", "pos", "edge_index", "batch", "ptr", "graph", "num_graphs", "ptr32", "seg_ptr"}
    for key, v in vars(samples[0]).items():
        if key in known or key.startswith("_"):
            continue
        if isinstance(v, (int, float)) or (torch.is_tensor(v) and v.dim() == 0):
            setattr(out, key, torch.tensor([float(getattr(smp, key)) if isinstance(v, float) else int(getattr(smp, key))
                                            for smp in samples]))
    return out


def merge_batches(batches: Sequence[Data]) -> Data:
    """Merged view of several task batches for the fused backbone pass (host side, before the move
    to the device): ``x`` stays a LIST of the per-task feature blocks (no copy of the 1536-d rows),
    positions / edges are concatenated with node offsets, and ``seg_ptr`` keeps the per-task row
    segments so that graph-LayerNorm statistics stay per task batch, as in the reference where every
    task batch is a separate backbone call (main_temporal.py:87-90)."""
    if batches and all(b.__dict__.get("_arena") is not None and b.__dict__["_arena"].layout.kind == "task" for b in batches):
        return _merge_arenas(batches)
    poss, eis, seg = [], [], [0]
    off = 0
    for b in batches:
        poss.append(b.pos)
        eis.append(b.edge_index + off)
        off += b.pos.shape[0]
        seg.append(off)
    ei = torch.cat(eis, dim=1)
    xs = [b.x for b in batches]
    # if the task blocks already are consecutive row ranges of ONE buffer (see ``pack_features``) the merged
    # batch exposes that buffer: the first contraction then runs once at M = all nodes
    base = getattr(batches[0], "x_base", None)
    x = base if (base is not None and all(getattr(b, "x_base", None) is base for b in batches)
                 and base.shape[0] == off) else xs
    out = Data(x=x, pos=torch.cat(poss), edge_index=ei)
    # The task batches occupy disjoint, ascending node ranges, so the merged CSR is the CONCATENATION of the tasks' CSR
    # arrays with node / edge offsets (what the stable sorts of build_csr over the merged edge list would give, entry for
    # entry) -- O(E) copies instead of two sorts over all edges (9 ms of host time per step), whatever the edge lists are
    # (the LTA edge set changes with the labels of every batch).
    graphs = [getattr(b, "graph", None) for b in batches]
    out.graph = concat_csr(graphs) if all(g is not None for g in graphs) else build_csr(ei, off)
    out.seg_ptr = torch.tensor(seg, dtype=torch.int32)
    out.num_segments = len(batches)
    out.min_seg_rows = min(b - a for a, b in zip(seg, seg[1:]))  # (host integer: the shortest task batch, in rows)
    if all(getattr(b, "pos_range", None) is not None for b in batches):
        out.pos_range = (min(b.pos_range[0] for b in batches), max(b.pos_range[1] for b in batches))
    return out


def concat_csr(graphs: Sequence[CSRGraph]) -> CSRGraph:
    """CSR of the disjoint union of graphs whose node ranges follow each other (== build_csr of the offset edge lists)."""
    def ptr(name):
        parts, eoff = [], 0
        for g in graphs:
            rp = getattr(g, name)
            parts.append(rp[:-1] + eoff)
            eoff += int(rp[-1])
        parts.append(torch.tensor([eoff], dtype=torch.int32, device=parts[0].device))
        return torch.cat(parts).to(torch.int32)

    def ids(name):
        parts, noff = [], 0
        for g in graphs:
            v = getattr(g, name)
            parts.append((v if v is not None else torch.zeros(0, dtype=torch.int32)) + noff)
            noff += g.num_nodes
        return torch.cat(parts).to(torch.int32)

    def mode(listed, name):
        with_rows = [getattr(g, name) for g, l in zip(graphs, listed) if l is not None and l.numel()]
        return int(bool(with_rows) and all(m == 1 for m in with_rows))
    heavy, t_heavy = ids("heavy"), ids("t_heavy")
    band = torch.cat([g.band for g in graphs]) if all(g.band is not None for g in graphs) else None  # (codes are relative)
    rowptr_all, col_all = ptr("rowptr"), ids("col")
    return CSRGraph(rowptr_all, col_all, ptr("t_rowptr"), ids("t_col"), torch.cat([g.t_wgt for g in graphs]),
                    sum(g.num_nodes for g in graphs), heavy, t_heavy, mode([g.heavy for g in graphs], "heavy_mode"),
                    mode([g.t_heavy for g in graphs], "t_heavy_mode"), band)


class PinnedRing:
    """A few persistent page-locked staging buffers per (shape, dtype), handed out in turn.  Allocating 57 MB of pinned
    memory per training step cost ~70 ms per step in the loop (hipHostMalloc / hipHostFree synchronise with the device);
    a buffer is handed out again only after the event recorded behind its last host-to-device copy has completed."""

    def __init__(self, depth: int = 3):
        self.depth, self.slots, self.next = depth, {}, {}

    def get(self, shape, dtype) -> "tuple[torch.Tensor, list]":
        key = (tuple(shape), dtype)
        ring = self.slots.setdefault(key, [])
        i = self.next.get(key, 0)
        if len(ring) < self.depth:
            ring.append([torch.empty(shape, dtype=dtype, pin_memory=True), None])
            i = len(ring) - 1
        self.next[key] = (i + 1) % self.depth
        slot = ring[i]
        if slot[1] is not None:
            slot[1].synchronize()  # the copy that last read this buffer
            slot[1] = None
        return slot[0], slot


_pinned_ring = PinnedRing()
_NUMPY_DTYPES = frozenset((torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8, torch.bool, torch.float32,
                           torch.float64, torch.float16))


# Arrays whose length is the EDGE count of a batch.  The LTA edge set depends on the labels (n_forecast counts ``y[:, 0] >
# 0``, reference lta_temp_connectivity.py:47), so E differs by a few entries from batch to batch.  No kernel takes E: the
# gathers walk ``rowptr[i] .. rowptr[i + 1]``.  The static buffers of a captured step therefore hold these arrays at a
# CAPACITY (E rounded up to EDGE_BUCKET), a replay writes the first E entries, and signatures compare capacities.
EDGE_FIELDS = (".edge_index", ".graph.col", ".graph.t_col", ".graph.t_wgt")
EDGE_BUCKET = 1024


def edge_capacity(e: int) -> int:
    return (int(e) + EDGE_BUCKET - 1) // EDGE_BUCKET * EDGE_BUCKET


class BlobRef:
    """What the batches of ONE packed transfer share (``to_device_packed``): the device byte buffer every tensor of theirs is
    a view of, a signature of its layout (every tensor's offset / shape -- edge-sized arrays at their capacity -- / dtype, every
    plain scalar), and ``rebuild(buffer, static)``: the same batches as views of ANOTHER buffer of that layout (``static``:
    edge-sized arrays at their capacity shape -- the private buffers of a captured step).  Two transfers with equal
    signatures differ in tensor VALUES only: a captured step takes the next batch with ONE device-to-device copy of the
    buffer instead of one copy per tensor (engine.StepBase.train_step)."""

    def __init__(self, dev, gsig, rebuild):
        self.dev, self.gsig, self.rebuild = dev, gsig, rebuild
        self.names = None  # what the caller calls the batches of the transfer, in order (task names, "merged")


# ---- batches built INTO one byte buffer (the native builders) ---------------------------------------------------------------------
# ``to_device_packed`` walks ~60 tensors per step to lay them out in one transfer buffer, and the builders allocate those tensors
# one by one: together ~0.8 ms of interpreter time per step of the headline workload, which no training loop ever looks at (a
# replayed step copies the transfer's buffer as a whole).  The native builders therefore write a batch's arrays straight into ONE
# buffer at the offsets of an ``ArenaLayout`` (each array 256-byte aligned, edge-sized arrays in regions of their capacity, cleared
# behind E); the host batch is a ``LazyData`` whose tensors are views of that buffer, built only when somebody reads a field;
# ``merge_batches`` builds the merged batch the same way and ``to_device_packed`` moves such batches with ONE memcpy each.
_NP_OF = {torch.int64: "int64", torch.int32: "int32", torch.uint8: "uint8", torch.float32: "float32", torch.float64: "float64",
          torch.int16: "int16", torch.int8: "int8", torch.bool: "bool"}


class ArenaLayout:
    """Offsets of a batch's arrays in one buffer.  ``fields``: (name, shape, torch dtype, edge) -- ``edge``: the last axis is an
    edge count; the region then has ``edge_cap`` entries per row and the logical array is its first E."""

    def __init__(self, kind: str, fields, edge_cap: int):
        import numpy as np
        self.kind, self.edge_cap = kind, int(edge_cap)
        self.fields, self.offs, off = [], {}, 0
        for name, shape, dt, edge in fields:
            cap = self.edge_cap if edge else 0
            stored = (*shape[:-1], cap) if cap else tuple(shape)
            nbytes = int(np.prod(stored, dtype=np.int64)) * torch.empty(0, dtype=dt).element_size()
            self.fields.append((name, off, tuple(shape), dt, cap, nbytes))
            self.offs[name] = off
            off += (nbytes + 255) // 256 * 256
        self.total = max(off, 256)
        self.token = hash((kind, tuple((n, o, (*sh[:-1], c) if c else sh, str(dt)) for n, o, sh, dt, c, _ in self.fields)))

    def views(self, buf: torch.Tensor, base: int, E: int, static: bool = False, only=None) -> dict:
        """name -> tensor view of ``buf`` (uint8, host or device) for a batch whose arena starts at byte ``base``; edge-sized
        arrays as [..., :E] of their region (``static``: the whole region -- the private buffers of a captured step)."""
        out = {}
        for name, off, shape, dt, cap, nbytes in self.fields:
            if only is not None and name not in only:
                continue
            if not nbytes:
                out[name] = torch.empty((*shape[:-1], E) if cap else shape, dtype=dt, device=buf.device)
            elif cap:
                region = buf[base + off:base + off + nbytes].view(dt).view((*shape[:-1], cap))
                out[name] = region if static else region[..., :E]
            else:
                out[name] = buf[base + off:base + off + nbytes].view(dt).view(shape)
        return out


class Arena:
    """One host batch in one buffer: ``buf`` (numpy uint8), its layout, the edge count, and the plain values that are not
    arrays (``meta``: num_graphs, pos_range, heavy-row modes, ...; part of a transfer's signature)."""

    def __init__(self, buf, layout: ArenaLayout, E: int, meta: dict):
        self.buf, self.layout, self.E, self.meta = buf, layout, int(E), meta
        self.sig = tuple(sorted(meta.items()))

    def addr(self, name: str) -> int:
        return self.buf.ctypes.data + self.layout.offs[name]


_ARENA_EAGER = ("num_graphs", "pos_range", "live_ap", "num_segments", "min_seg_rows")


def _arena_fields(layout: ArenaLayout, meta: dict, buf: torch.Tensor, base: int, E: int, static: bool = False) -> dict:
    """The fields of a batch (task batch or merged batch) as views of ``buf``, in the order the tensor-by-tensor builders set them."""
    v = layout.views(buf, base, E, static)
    g = CSRGraph(v["graph.rowptr"], v["graph.col"], v["graph.t_rowptr"], v["graph.t_col"], v["graph.t_wgt"], meta["num_nodes"],
                 v["graph.heavy"], v["graph.t_heavy"], meta["heavy_mode"], meta["t_heavy_mode"], v["graph.band"])
    if layout.kind == "merged":
        out = dict(x=None, pos=v["pos"], edge_index=v["edge_index"], graph=g, seg_ptr=v["seg_ptr"],
                   num_segments=meta["num_segments"], min_seg_rows=meta["min_seg_rows"])
        if "pos_range" in meta:
            out["pos_range"] = meta["pos_range"]
        return out
    out = dict(x=None, y=v["y"], pos=v["pos"], edge_index=v["edge_index"], batch=v["batch"], ptr=v["ptr"], num_graphs=meta["num_graphs"])
    if "pos_range" in meta:
        out["pos_range"] = meta["pos_range"]
    out.update(x_idx=v["x_idx"], graph=g, ptr32=v["ptr32"], seg_ptr=v["seg_ptr"])
    if "live_idx" in v:
        out.update(live_idx=v["live_idx"], live_inv=v["live_inv"], live_y=v["live_y"])
        if "live_ap" in meta:
            out["live_ap"] = meta["live_ap"]
    for name in v:
        if name.startswith("attr."):
            out[name[5:]] = v[name]
    return out


def arena_batch(arena: Arena) -> "LazyData":
    """The host batch of an arena: fields are views of its buffer, built on first use."""
    host = torch.from_numpy(arena.buf)
    out = LazyData(lambda: _arena_fields(arena.layout, arena.meta, host, 0, arena.E))
    d = out.__dict__
    d["_arena"] = arena
    d["x"] = None
    for k in _ARENA_EAGER:
        if k in arena.meta:
            d[k] = arena.meta[k]
    if arena.layout.kind == "task":  # (the staging path reads the store rows of every step: one view, not the batch's twenty)
        d["x_idx"] = arena.layout.views(host, 0, arena.E, only=("x_idx",))["x_idx"]
    return out


def _merge_arenas(batches: Sequence["Data"]) -> "Data":
    """``merge_batches`` for batches that live in arenas: ONE host call (egk_host_merge_batches) writes the merged batch's arena."""
    import ctypes as C
    import numpy as np
    from . import _lib
    ars = [b._arena for b in batches]
    k = len(ars)
    parts = (_lib.HostPart * k)()
    N = E = nh = nth = 0
    for p, a in zip(parts, ars):
        m, lay = a.meta, a.layout
        p.n_nodes, p.E, p.n_heavy, p.n_t_heavy = m["num_nodes"], a.E, m["n_heavy"], m["n_t_heavy"]
        p.edge_stride = lay.edge_cap or a.E
        p.pos_min, p.pos_max = m["pos_range"]
        p.heavy_mode, p.t_heavy_mode = m["heavy_mode"], m["t_heavy_mode"]
        base = a.buf.ctypes.data
        for name, fld in _PART_PTRS:
            setattr(p, fld, base + lay.offs[name])
        N, E, nh, nth = N + p.n_nodes, E + a.E, nh + p.n_heavy, nth + p.n_t_heavy
    key = (k, N, edge_capacity(E), nh, nth)
    lay = _merged_layouts.get(key)
    if lay is None:
        i64, i32 = torch.int64, torch.int32
        lay = _merged_layouts[key] = ArenaLayout("merged", [
            ("pos", (N,), i64, False), ("edge_index", (2, E), i64, True), ("graph.rowptr", (N + 1,), i32, False),
            ("graph.col", (E,), i32, True), ("graph.t_rowptr", (N + 1,), i32, False), ("graph.t_col", (E,), i32, True),
            ("graph.t_wgt", (E,), torch.float32, True), ("graph.heavy", (nh,), i32, False), ("graph.t_heavy", (nth,), i32, False),
            ("graph.band", (N,), torch.uint8, False), ("seg_ptr", (k + 1,), i32, False)], edge_capacity(E))
    buf = np.empty(lay.total, dtype=np.uint8)
    base = buf.ctypes.data
    om = _lib.HostMerged(edge_cap=lay.edge_cap, **{fld: base + lay.offs[name] for name, fld in _MERGED_PTRS})
    rc = int(_lib.load().egk_host_merge_batches(parts, k, C.byref(om)))
    if rc != 0:
        raise ValueError(f"egk_host_merge_batches failed (code {rc})")
    meta = dict(num_nodes=N, n_heavy=nh, n_t_heavy=nth, heavy_mode=int(om.heavy_mode), t_heavy_mode=int(om.t_heavy_mode),
                num_segments=k, min_seg_rows=int(om.min_seg_rows), pos_range=(int(om.pos_min), int(om.pos_max)))
    out = arena_batch(Arena(buf, lay, E, meta))
    xs = [b.__dict__.get("x") for b in batches]
    base_x = batches[0].__dict__.get("x_base")
    out.x = base_x if (base_x is not None and all(b.__dict__.get("x_base") is base_x for b in batches)
                       and base_x.shape[0] == N) else xs
    return out


_merged_layouts = {}
_BATCH_PTRS = (("y", "y"), ("pos", "pos"), ("batch", "batch"), ("ptr", "ptr"), ("ptr32", "ptr32"), ("x_idx", "x_idx"),
               ("edge_index", "edge_index"), ("graph.rowptr", "rowptr"), ("graph.col", "col"), ("graph.t_rowptr", "t_rowptr"),
               ("graph.t_col", "t_col"), ("graph.t_wgt", "t_wgt"), ("graph.band", "band"), ("graph.heavy", "heavy"),
               ("graph.t_heavy", "t_heavy"))
_BATCH_PTRS_LIVE = (*_BATCH_PTRS, ("live_idx", "live_idx"), ("live_inv", "live_inv"), ("live_y", "live_y"))
_PART_PTRS = (("pos", "pos"), ("edge_index", "edge_index"), ("graph.rowptr", "rowptr"), ("graph.col", "col"),
              ("graph.t_rowptr", "t_rowptr"), ("graph.t_col", "t_col"), ("graph.t_wgt", "t_wgt"), ("graph.band", "band"),
              ("graph.heavy", "heavy"), ("graph.t_heavy", "t_heavy"))
_MERGED_PTRS = (*_PART_PTRS, ("seg_ptr", "seg_ptr"))


def _pack_arenas(datas, device, non_blocking: bool, on_gpu: bool) -> List["Data"]:
    """``to_device_packed`` for batches that live in arenas: one memcpy per batch into the page-locked transfer buffer, one
    host-to-device copy, lazy views of the device buffer."""
    ars = [d._arena for d in datas]
    bases, total = [], 0
    for a in ars:
        bases.append(total)
        total += (a.layout.total + 255) // 256 * 256
    size = (total + 65535) // 65536 * 65536
    if on_gpu:
        host, slot = _pinned_ring.get((size,), torch.uint8)
    else:
        host, slot = torch.empty(size, dtype=torch.uint8), None
    hn = host.numpy()
    for a, b in zip(ars, bases):
        hn[b:b + a.layout.total] = a.buf
    if on_gpu:
        dev = host.to(device, non_blocking=non_blocking)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        slot[1] = ev
    else:
        dev = host.clone()
    specs = [(a.layout, a.meta, b, a.E) for a, b in zip(ars, bases)]  # (the host buffers are not kept alive by the transfer)

    def rebuild(buf, static: bool = False, lazy: bool = False):
        outs = []
        for lay, meta, b, E in specs:
            if lazy:
                ld = LazyData(lambda lay=lay, meta=meta, b=b, E=E: _arena_fields(lay, meta, buf, b, E, static))
                ld.__dict__["x"] = None
                for k in _ARENA_EAGER:
                    if k in meta:
                        ld.__dict__[k] = meta[k]
                if lay.kind == "task":
                    ld.__dict__["x_idx"] = lay.views(buf, b, E, static, only=("x_idx",))["x_idx"]
                outs.append(ld)
            else:
                outs.append(Data(**_arena_fields(lay, meta, buf, b, E, static)))
        return outs
    outs = rebuild(dev, lazy=True)
    ref = BlobRef(dev, tuple((a.layout.token, b, a.sig) for a, b in zip(ars, bases)), rebuild)
    for o in outs:
        o._blob = ref
    return outs


def to_device_packed(datas: Sequence["Data"], device, non_blocking: bool = True, pack_on_cpu: bool = False) -> List["Data"]:
    """``[d.to(device) for d in datas]`` with ONE host-to-device copy: every tensor of the batches (labels, positions,
    CSR arrays, per-sample attributes ...) is laid out in one page-locked byte buffer of ``_pinned_ring`` and the device
    tensors are views of its device copy.  A step's ~55 small tensors cost ~0.16 ms EACH as separate pageable copies
    (9 ms per step); packed, they are one 1-2 MB transfer.  Edge-sized arrays get a region of their CAPACITY (their length
    rounded up to EDGE_BUCKET entries), so that steps whose edge counts differ by a few entries share one layout; every
    returned batch carries the transfer's ``BlobRef`` as ``_blob``.  (``pack_on_cpu``: the same layout with a CPU target, for
    the tests of the layout.)"""
    from dataclasses import fields, is_dataclass, replace
    on_gpu = torch.device(device).type == "cuda"
    if not on_gpu and not pack_on_cpu:
        return [d.to(device, non_blocking=non_blocking) for d in datas]
    if datas and all(d.__dict__.get("_arena") is not None for d in datas):
        return _pack_arenas(datas, device, non_blocking, on_gpu)
    items = []  # (tensor, byte offset, capacity of the last axis or 0)
    total = 0
    sig = []

    def plan(t, path):
        nonlocal total
        if t.device.type != "cpu":
            sig.append((path, "device"))
            return None
        cap = edge_capacity(t.shape[-1]) if (t.dim() >= 1 and path.endswith(EDGE_FIELDS)) else 0
        numel = (t.numel() // max(t.shape[-1], 1)) * cap if cap else t.numel()
        off = total
        total += (numel * t.element_size() + 255) // 256 * 256
        items.append((t, off, cap))
        sig.append((path, off, tuple(t.shape[:-1]) + (cap,) if cap else tuple(t.shape), t.dtype))
        return (off, cap)

    def walk(v, path):
        if torch.is_tensor(v):
            return ("t", v, plan(v, path))
        if is_dataclass(v):
            return ("dc", v, {f.name: walk(getattr(v, f.name), f"{path}.{f.name}") for f in fields(v)})
        if isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
            return ("l", v, [walk(t, f"{path}[{i}]") for i, t in enumerate(v)])
        if isinstance(v, (int, float, bool, str)) and "._" not in path:
            sig.append((path, v))
        elif (isinstance(v, (list, tuple)) and "._" not in path
              and all(isinstance(e, (int, float, bool, str)) for e in v)):
            # tuples of plain scalars (pos_range = (min, max) ...) are baked into a captured step's kernel arguments like
            # the scalars themselves (engine.batch_signature descends into them): part of the layout signature too
            sig.append((path, tuple(v)))
        return ("o", v, None)
    plans = [{k: walk(v, f"{i}.{k}") for k, v in d.materialise().__dict__.items() if k != "_arena"} for i, d in enumerate(datas)]
    if total == 0:
        return [d.to(device, non_blocking=non_blocking) for d in datas]
    # (staging size rounded up to 64 KiB: batches whose edge counts differ by a few entries reuse one ring of buffers
    #  instead of allocating a new page-locked ring per distinct byte total)
    if on_gpu:
        host, slot = _pinned_ring.get(((total + 65535) // 65536 * 65536,), torch.uint8)
    else:
        host, slot = torch.empty((total + 65535) // 65536 * 65536, dtype=torch.uint8), None

    def view_of(buf, t, off, cap, full):
        """The tensor's view of ``buf``: [..., :E] of its [..., capacity] region for an edge-sized array (``full``: the region)."""
        if cap:
            lead = tuple(t.shape[:-1])
            n = (t.numel() // max(t.shape[-1], 1)) * cap * t.element_size()
            region = buf[off:off + n].view(t.dtype).view(lead + (cap,))
            return region if full else region[..., :t.shape[-1]]
        n = t.numel() * t.element_size()
        return buf[off:off + n].view(t.dtype).view(t.shape) if n else torch.empty(t.shape, dtype=t.dtype, device=buf.device)

    hn = host.numpy()
    for t, off, cap in items:  # (numpy slices: a third of the cost of torch views for ~60 small copies)
        a = t.numpy() if (t.is_contiguous() and t.dtype in _NUMPY_DTYPES) else None
        if a is None:
            if t.numel():
                view_of(host, t, off, cap, False).copy_(t)
            if cap and cap != t.shape[-1]:
                view_of(host, t, off, cap, True)[..., t.shape[-1]:].zero_()  # (entries no row range reaches)
        elif cap:
            e = a.shape[-1]
            lead = a.size // max(e, 1) if e else int(np.prod(a.shape[:-1], dtype=np.int64))
            region = hn[off:off + lead * cap * a.itemsize].view(a.dtype).reshape(lead, cap)
            region[:, :e] = a.reshape(lead, e)
            region[:, e:] = 0
        elif a.size:
            hn[off:off + a.size * a.itemsize] = a.reshape(-1).view(np.uint8)
    if on_gpu:
        dev = host.to(device, non_blocking=non_blocking)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        slot[1] = ev
    else:
        dev = host.clone()

    def build(p, buf, full):
        kind, v, extra = p
        if kind == "t":
            if extra is None:
                return v.to(device, non_blocking=non_blocking)
            return view_of(buf, v, extra[0], extra[1], full)
        if kind == "dc":
            return replace(v, **{name: build(q, buf, full) for name, q in extra.items()})
        if kind == "l":
            return [build(q, buf, full) for q in extra]
        return v

    def rebuild(buf, static: bool = False, lazy: bool = False):
        if lazy:
            lazies = []
            for pl in plans:
                ld = LazyData(lambda pl=pl: {k: build(q, buf, static) for k, q in pl.items()})
                # plain host attributes (num_graphs, pos_range, heavy-row modes ...) are there at once: a training loop that
                # only counts sequences (main_temporal / main_egopack: ``int(b.num_graphs)``) must not trigger the ~20 views
                for k, (kind, v, _) in pl.items():
                    if kind == "o":
                        ld.__dict__[k] = v
                lazies.append(ld)
            return lazies
        outs = []
        for pl in plans:
            out = Data()
            for k, q in pl.items():
                setattr(out, k, build(q, buf, static))
            outs.append(out)
        return outs
    outs = rebuild(dev, lazy=True)
    # (a tensor that was on a device already is not part of the buffer: no layout signature, the per-tensor paths apply)
    ref = BlobRef(dev, None if any(e[1:] == ("device",) for e in sig) else tuple(sig), rebuild)
    for o in outs:
        o._blob = ref
    return outs


def pack_features(batches: Sequence[Data], dtype=None, device=None, pin: bool = False, ring_slot: list = None) -> torch.Tensor:
    """Copy the feature blocks of several task batches into ONE [sum N, S, F] buffer (one host->device
    transfer, one merged contraction) and re-point every ``batch.x`` at its row range of that buffer.
    ``pin``: the buffer is one of the persistent page-locked buffers of ``_pinned_ring``; the caller that copies it to a
    device records an event behind that copy in ``ring_slot[1]`` (pass a list to receive the slot)."""
    n = sum(b.x.shape[0] for b in batches)
    ref = batches[0].x
    shape, dt = (n, *ref.shape[1:]), dtype or ref.dtype
    if pin and (device is None or str(device) == "cpu") and torch.cuda.is_available():
        buf, slot = _pinned_ring.get(shape, dt)
        if ring_slot is not None:
            ring_slot.append(slot)
    else:
        buf = torch.empty(shape, dtype=dt, device=device or ref.device)
    off = 0
    for b in batches:
        m = b.x.shape[0]
        buf[off:off + m].copy_(b.x)
        b.x = buf[off:off + m]
        b.x_base = buf
        off += m
    return buf


# --------------------------------------------------------------------------------------------
# loaders
# --------------------------------------------------------------------------------------------
def multiloader(loaders, weights):
    """One tuple of batches per step from several loaders walked side by side (reference utils/dataloading.py:8-47): slot i is
    ``None`` for a loader that is None or has weight <= 0; a loader that runs out is started again -- its first batch takes the
    slot -- until EVERY active loader has run out at least once; the step in which the last of them runs out ends the walk (what the
    earlier slots of that step had drawn is dropped, as in the reference).  A generator: the state is its frame."""
    active = [i for i, (ld, w) in enumerate(zip(loaders, weights)) if ld is not None and w > 0]
    walks = {i: iter(loaders[i]) for i in active}
    pending = set(active)  # loaders that have not run out yet
    while pending:  # (no active loader at all: nothing to walk -- the reference's class would hand out empty steps for ever)
        step = [None] * len(loaders)
        for i in active:
            batch = next(walks[i], _EXHAUSTED)
            if batch is _EXHAUSTED:
                pending.discard(i)
                if not pending:
                    return
                walks[i] = iter(loaders[i])
                batch = next(walks[i])
            step[i] = batch
        yield tuple(step)


_EXHAUSTED = object()


class BatchLoader:
    """Minimal seeded loader: shuffles sample indices with a torch.Generator (re-drawn per epoch),
    shards them by rank, collates ``batch_size`` samples.
    Stands in for utils/dataloading.build_dataloader + PyG DataLoader (reference :56-70).

    shard="samples" (training): rank-strided over the common permutation, equal share per rank.
    shard="batches" (evaluation passes): every rank cuts the SAME batches the single-process loader would and takes
    batches rank, rank+world, ... -- the graph-mode LayerNorm statistics span a whole batch, so only
    identical batches give the single-process logits; ranks may get one batch more or less than each other, which is
    fine for passes without a per-step collective."""

    def __init__(self, dataset, batch_size: int, shuffle: bool, drop_last: bool, seed: int = 0, rank: int = 0,
                 world_size: int = 1, pin_memory: bool = False, shard: str = "samples"):
        if shard not in ("samples", "batches"):
            raise ValueError(f"shard={shard!r}: 'samples' or 'batches'")
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.rank, self.world_size, self.pin_memory, self.shard = rank, world_size, pin_memory, shard
        self.seed = int(seed)
        self.gen = torch.Generator()
        self.gen.manual_seed(seed)

    def _indices(self) -> List[int]:
        n = len(self.dataset)
        idx = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
        if self.world_size > 1 and self.shard == "samples":  # equal share per rank: same number of steps everywhere
            per = n // self.world_size
            idx = idx[: per * self.world_size][self.rank:: self.world_size]
        return idx

    def __len__(self):
        if self.world_size > 1 and self.shard == "batches":
            n = len(self.dataset)
            total = n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)
            return len(range(self.rank, total, self.world_size))
        n = len(self.dataset) // self.world_size if self.world_size > 1 else len(self.dataset)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def state_dict(self) -> dict:
        """Shuffle generator state: a resumed run draws the permutations the uninterrupted run would have drawn."""
        return {"generator": self.gen.get_state()}

    def load_state_dict(self, state: dict) -> None:
        self.gen.set_state(state["generator"].cpu())  # (a checkpoint loaded with map_location=device moved it)

    def _chunks(self):
        idx = self._indices()
        by_batch = self.world_size > 1 and self.shard == "batches"
        for b, i in enumerate(range(0, len(idx), self.batch_size)):
            chunk = idx[i: i + self.batch_size]
            if len(chunk) < self.batch_size and self.drop_last:
                return
            if by_batch and b % self.world_size != self.rank:
                continue
            yield chunk

    def __iter__(self):
        if self.workers > 0:
            yield from self._iter_workers()
            return
        for chunk in self._chunks():
            b = collate_chunk(self.dataset, chunk)
            yield b.pin_memory() if self.pin_memory else b

    # -- worker processes (opt-in: ``workers`` > 0) ------------------------------------------------------------------
    # Building a sample is per-sample Python (the reference's segment-sampling index arithmetic, edge builders): ~1 ms per
    # sample, 100-200 ms per step of 192 samples in the training process against a 1.8 ms device step.  ``workers``
    # processes (they never touch the device) collate whole batches ahead, at most 2 per worker in flight, and hand
    # them back in order.  Datasets whose samples depend on a sequential random stream (segment sampling with a shared
    # RandomState) draw every chunk from a stream seeded by (loader seed, epoch, chunk index): still the reference's
    # sampling, reproducible from the seed, not the single-process sequence.
    workers = 0
    _pool = None

    def start_workers(self):
        """Create the collation processes.  Entry points call this BEFORE the process touches the GPU (plain ``fork`` of a
        process without HIP state); if the device is already initialised the pool comes from a ``forkserver`` (a clean
        helper process forks the workers: no HIP runtime locks, threads or pinned mappings are inherited)."""
        import torch.multiprocessing as mp
        if self._pool is None and self.workers > 0:
            prep = getattr(self.dataset, "_tables", None)
            if prep is not None:
                prep()  # (the whole-batch builder's tables: built once here, shared copy-on-write by the forked workers)
            ctx = mp.get_context("forkserver" if torch.cuda.is_initialized() else "fork")
            self._pool = ctx.Pool(self.workers, initializer=_worker_init, initargs=(self.dataset,))
        return self

    def _iter_workers(self):
        import collections
        self.start_workers()
        self._epoch = getattr(self, "_epoch", -1) + 1
        pending = collections.deque()
        try:
            for n, chunk in enumerate(self._chunks()):
                # the sampling stream of a chunk is a function of (loader seed, epoch, chunk index): the batches do not
                # depend on which worker builds them
                pending.append(self._pool.apply_async(_worker_collate, (chunk, (self.seed, self._epoch, n))))
                if len(pending) >= 2 * self.workers:
                    b = unpack_data(*pending.popleft().get(timeout=300))
                    yield b.pin_memory() if self.pin_memory else b
            while pending:
                b = unpack_data(*pending.popleft().get(timeout=300))
                yield b.pin_memory() if self.pin_memory else b
        finally:
            if pending:  # the consumer stopped early: drop the in-flight work with its processes (a new pool next time)
                self.close()

    def close(self):
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_worker_dataset = None


def _worker_init(dataset):
    global _worker_dataset
    _worker_dataset = dataset
    torch.set_num_threads(1)


def _worker_collate(chunk, stream=None):
    rng = getattr(_worker_dataset, "rng", None)  # datasets that sample with a shared RandomState: one stream per chunk
    if stream is not None and rng is not None and hasattr(rng, "seed"):
        seed, epoch, n = stream
        rng.seed((int(seed) * 1000003 + int(epoch) * 7919 + int(n) * 104729 + int(getattr(_worker_dataset, "seed", 0))) % (2 ** 32))
    return pack_data(collate_chunk(_worker_dataset, chunk))


def pack_data(d: "Data"):
    """(ONE uint8 tensor, layout) holding every tensor of a host batch (CSR arrays included): what a collation process hands
    back.  A batch is ~25 small tensors; sent one by one through torch's shared-memory pickling each costs a file descriptor
    and ~0.3 ms (15 ms per step measured with three loaders) -- packed, a batch is one shared buffer plus a small tuple."""
    from dataclasses import fields, is_dataclass
    items, total = [], 0

    def plan(t):
        nonlocal total
        off = total
        total += (t.numel() * t.element_size() + 63) // 64 * 64
        items.append((t, off))
        return ("t", off, tuple(t.shape), t.dtype)

    def walk(v):
        if torch.is_tensor(v):
            return plan(v)
        if is_dataclass(v):
            return ("dc", type(v), {f.name: walk(getattr(v, f.name)) for f in fields(v)})
        if isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
            return ("l", [plan(t) for t in v])
        return ("o", v)
    spec = {k: walk(v) for k, v in d.materialise().__dict__.items() if k != "_arena"}
    buf = torch.empty(max(total, 1), dtype=torch.uint8)
    for t, off in items:
        n = t.numel() * t.element_size()
        if n:
            buf[off:off + n].view(t.dtype).view(t.shape).copy_(t)
    return buf, spec


def unpack_data(buf: torch.Tensor, spec) -> "Data":
    """The batch of ``pack_data``: its tensors are views of ``buf`` (no copy)."""
    def build(p):
        if p[0] == "t":
            _, off, shape, dtype = p
            n = 1
            for s_ in shape:
                n *= s_
            nb = n * torch.empty(0, dtype=dtype).element_size()
            return buf[off:off + nb].view(dtype).view(shape) if nb else torch.empty(shape, dtype=dtype)
        if p[0] == "dc":
            return p[1](**{name: build(q) for name, q in p[2].items()})
        if p[0] == "l":
            return [build(q) for q in p[1]]
        return p[1]
    out = Data()
    for k, p in spec.items():
        setattr(out, k, build(p))
    return out


def collate_chunk(dataset, chunk) -> Data:
    """One batch of ``dataset``: its whole-batch builder when it has one (``batch(chunk)``: vector arithmetic, same result
    and same random-stream consumption as sample-by-sample collation), else ``collate`` of the samples."""
    build = getattr(dataset, "batch", None)
    if build is not None and getattr(dataset, "vectorised_batches", True):
        return build(chunk)
    return collate([dataset[j] for j in chunk])


def build_dataloader(dataset, batch_size, shuffle, num_workers, drop_last, seed=0, rank=0, world_size=1, shard="samples",
                     workers: int = 0):
    """Signature of the reference's build_dataloader.  ``num_workers`` (the reference's DataLoader argument) is accepted
    and collation stays in-process unless ``workers`` > 0 asks for forked collation processes (BatchLoader)."""
    dl = BatchLoader(dataset, batch_size, shuffle, drop_last, seed, rank, world_size, shard=shard)
    dl.workers = int(workers)
    return dl


# --------------------------------------------------------------------------------------------
# synthetic Omnivore-shaped datasets: no Ego4D data is available or needed
# --------------------------------------------------------------------------------------------
class SyntheticTaskDataset:
    """Deterministic synthetic samples with the shapes / label conventions of the Ego4D datasets:
    x ~ N(0,1) [T, S, F]; AR: only the centre node labelled; LTA: first 2 nodes unlabelled inputs;
    OSCC: one label per sequence; PNR: one positive node per sequence."""

    has_joint_label = False
    num_labels = 2
    label_names = ["verbs", "nouns"]

    def __init__(self, task: str, length: int, T: int, num_segments: int = 3, features_size: int = 1536,
                 num_class_labels=(115, 478), k: int = 1, seed: int = 1, transform=None):
        self.task, self.length, self.T, self.S, self.features_size = task, length, T, num_segments, features_size
        self.num_class_labels, self.seed = tuple(num_class_labels), seed
        self.class_labels = [[f"verb_{i}" for i in range(self.num_class_labels[0])],
                             [f"noun_{i}" for i in range(self.num_class_labels[1])]]
        self.lta_nodes = T  # the LTA meter cuts the label vector into sequences of this many nodes (22 on Ego4D)
        if transform is None:
            transform = LTATemporalConnectivity(r=k + 0.5) if task == "lta" else RadiusGraph(r=k + 0.5)
        self.transform = transform

    def __len__(self):
        return self.length

    def __getitem__(self, i: int, with_x: bool = True) -> Data:
        g = torch.Generator()
        g.manual_seed(self.seed * 1000003 + i)
        T, V, Nn = self.T, *self.num_class_labels
        # (``with_x=False``: labels / positions / edges only -- the resident dataset indexes its features elsewhere, and
        #  147 k normal deviates per sample were 150 ms of host time per step of 192 samples)
        x = torch.randn(T, self.S, self.features_size, generator=g) if with_x else None
        if self.task == "ar":
            pos = torch.arange(T) - T // 2
            y = torch.full((T, 2), -1, dtype=torch.long)
            y[T // 2, 0] = torch.randint(0, V, (1,), generator=g)
            y[T // 2, 1] = torch.randint(0, Nn, (1,), generator=g)
        elif self.task == "lta":
            pos = torch.arange(T)
            y = torch.stack([torch.randint(0, V, (T,), generator=g), torch.randint(0, Nn, (T,), generator=g)], 1)
            y[:2] = -1
        elif self.task == "oscc":
            pos = torch.arange(T)
            y = int(torch.randint(0, 2, (1,), generator=g))
        elif self.task == "pnr":
            pos = torch.arange(T)
            y = torch.zeros(T, dtype=torch.long)
            at = int(torch.randint(0, T, (1,), generator=g))
            y[at] = 1
            # clip of 8 s at 30 fps starting at a random frame; the PNR frame sits inside the positive node's span
            start = int(torch.randint(0, 10000, (1,), generator=g))
            d = self.transform(Data(x=x, pos=pos, y=y, batch=None))
            d.start_frame, d.end_frame = start, start + 240
            d.pnr_frame = start + int((at + 0.5) * 240 / T)
            return d
        else:
            raise ValueError(self.task)
        return self.transform(Data(x=x, pos=pos, y=y, batch=None))


class LearnableSyntheticDataset(SyntheticTaskDataset):
    """Synthetic samples whose labels can be LEARNED from the features (the plain synthetic datasets carry random labels:
    fine for throughput, useless for a metric).  A fixed code book (one F-vector per verb / noun / event, the same for
    every split) is added to the N(0,1) features with amplitude ``signal``:
      AR   the centre node's (verb, noun) codes on the centre node and its two neighbours;
      LTA  one (verb, noun) per sequence, every forecast node is labelled with it: its codes at full amplitude on the 2
           observed nodes and at half amplitude on the forecast nodes;
      OSCC label 1 = one random node carries the state-change code;
      PNR  the positive node carries the point-of-no-return code.
    Used by the fixed-seed metric-agreement test (f32 vs bf16 training) and as a sanity workload."""

    def __init__(self, task: str, length: int, T: int, num_segments: int = 3, features_size: int = 1536,
                 num_class_labels=(115, 478), k: int = 1, seed: int = 1, transform=None, signal: float = 1.0,
                 code_seed: int = 4242):
        super().__init__(task, length, T, num_segments, features_size, num_class_labels, k, seed, transform)
        g = torch.Generator().manual_seed(code_seed)  # the code book does not depend on the split's sample seed
        V, Nn = self.num_class_labels
        self.codes_v = torch.randn(V, features_size, generator=g)
        self.codes_n = torch.randn(Nn, features_size, generator=g)
        self.code_event = torch.randn(2, features_size, generator=g)  # [0]: OSCC state change, [1]: PNR
        self.signal = float(signal)

    def __getitem__(self, i: int, with_x: bool = True) -> Data:
        d = SyntheticTaskDataset.__getitem__(self, i)
        g = torch.Generator().manual_seed(self.seed * 7_000_003 + i)
        T, s = self.T, self.signal
        x = d.x
        if self.task == "ar":
            c = T // 2
            v, n = int(d.y[c, 0]), int(d.y[c, 1])
            x[max(c - 1, 0): c + 2] += s * (self.codes_v[v] + self.codes_n[n])
        elif self.task == "lta":
            V, Nn = self.num_class_labels
            v, n = int(torch.randint(1, V, (1,), generator=g)), int(torch.randint(0, Nn, (1,), generator=g))
            d.y[2:, 0], d.y[2:, 1] = v, n
            x[:2] += s * (self.codes_v[v] + self.codes_n[n])
            x[2:] += 0.5 * s * (self.codes_v[v] + self.codes_n[n])
            d = self.transform(Data(x=x, pos=d.pos, y=d.y, batch=None))  # (the LTA edges count the forecast labels)
        elif self.task == "oscc":
            if int(d.y) == 1:
                x[int(torch.randint(0, T, (1,), generator=g))] += s * self.code_event[0]
        elif self.task == "pnr":
            x[int(torch.argmax(d.y))] += s * self.code_event[1]
        d.x = x
        return d


# --------------------------------------------------------------------------------------------
# whole-batch builders (the input pipeline at device speed)
# --------------------------------------------------------------------------------------------
# Per-sample Python (one Data per sample, one edge transform, T sampling calls, then concatenation and two sorts for the CSR)
# costs ~1 ms per sample: 100-200 ms per step of 192 samples against a 1.5 ms device step.  A batch is instead assembled with
# vector arithmetic: segment sampling for all B x T windows at once on the reference's random stream
# (feature_store.window_rows_batch), labels / positions by index selection, and the graph structure from per-sample TEMPLATES --
# a temporal radius graph depends on the sequence length only, the LTA connectivity on the number of forecast labels -- whose
# CSR arrays are laid side by side with node / edge offsets (what the stable sorts of build_csr give for disjoint node ranges).
def _ragged(table, counts, tau, sample_of=None, pos_in=None):
    """Concatenation of ``table[tau[b], ..., :counts[tau[b]]]`` over b (``table`` padded along its last axis)."""
    import numpy as np
    c = counts[tau]
    if sample_of is None:
        off = np.cumsum(c) - c
        sample_of = np.repeat(np.arange(tau.shape[0]), c)
        pos_in = np.arange(int(c.sum())) - off[sample_of]
    return table[tau[sample_of], ..., pos_in], sample_of, pos_in


class GraphTemplates:
    """Distinct per-sample edge lists of a dataset (keyed by content) with their CSR arrays, padded into tables so that a
    batch's graph is a handful of index operations (``assemble``)."""

    def __init__(self, T: int):
        self.T, self.keys, self.items, self._tables = int(T), {}, [], None

    def add(self, edge_index: torch.Tensor) -> int:
        key = edge_index.numpy().tobytes()
        tid = self.keys.get(key)
        if tid is None:
            g = build_csr(edge_index, self.T)
            tid = self.keys[key] = len(self.items)
            self.items.append((edge_index.numpy().astype("int64"), g))
            self._tables = None
        return tid

    def tables(self):
        import numpy as np
        if self._tables is None:
            n, T = len(self.items), self.T
            e = np.array([ei.shape[1] for ei, _ in self.items], dtype=np.int64)
            em = max(int(e.max()), 1)
            EI = np.zeros((n, 2, em), dtype=np.int64)
            COL, TCOL, TW = np.zeros((n, em), dtype=np.int64), np.zeros((n, em), dtype=np.int64), np.zeros((n, em), dtype=np.float32)
            RP, TRP, BAND = np.zeros((n, T + 1), dtype=np.int64), np.zeros((n, T + 1), dtype=np.int64), np.zeros((n, T), dtype=np.uint8)
            nh = np.array([g.heavy.numel() for _, g in self.items], dtype=np.int64)
            nth = np.array([g.t_heavy.numel() for _, g in self.items], dtype=np.int64)
            HV, THV = np.zeros((n, max(int(nh.max()), 1)), dtype=np.int64), np.zeros((n, max(int(nth.max()), 1)), dtype=np.int64)
            dmax, tdmax = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
            for i, (ei, g) in enumerate(self.items):
                k = ei.shape[1]
                EI[i, :, :k] = ei
                COL[i, :k], TCOL[i, :k], TW[i, :k] = g.col.numpy(), g.t_col.numpy(), g.t_wgt.numpy()
                RP[i], TRP[i], BAND[i] = g.rowptr.numpy(), g.t_rowptr.numpy(), g.band.numpy()
                HV[i, :nh[i]], THV[i, :nth[i]] = g.heavy.numpy(), g.t_heavy.numpy()
                dmax[i] = int(np.diff(RP[i]).max()) if T else 0
                tdmax[i] = int(np.diff(TRP[i]).max()) if T else 0
            self._tables = dict(e=e, EI=EI, COL=COL, TCOL=TCOL, TW=TW, RP=RP, TRP=TRP, BAND=BAND, nh=nh, nth=nth, HV=HV, THV=THV,
                                dmax=dmax, tdmax=tdmax)
        return self._tables

    def assemble(self, tau):
        """(edge_index [2, E] int64, CSRGraph) of the batch whose b-th sample has template ``tau[b]`` == what ``collate`` builds
        from the samples' edge lists (offset-and-concatenate, then build_csr)."""
        import numpy as np
        t, T = self.tables(), self.T
        tau = np.asarray(tau, dtype=np.int64)
        B = tau.shape[0]
        # batches of one template sequence share their structure (every batch of a radius-graph task does; LTA batches
        # differ by their forecast-label counts): the assembled arrays are kept per sequence, read-only by convention
        key = tau.tobytes()
        hit = self._assembled.get(key) if hasattr(self, "_assembled") else None
        if hit is not None:
            return hit
        if not hasattr(self, "_assembled"):
            self._assembled = {}
        e = t["e"][tau]
        eoff, noff = np.cumsum(e) - e, np.arange(B, dtype=np.int64) * T
        E = int(e.sum())
        ei, sample_of, pos_in = _ragged(t["EI"], t["e"], tau)
        shift = noff[sample_of]
        ei = (ei.T if ei.ndim == 2 else ei.reshape(2, 0)) + shift[None, :]
        col = t["COL"][tau[sample_of], pos_in] + shift
        t_col = t["TCOL"][tau[sample_of], pos_in] + shift
        t_wgt = t["TW"][tau[sample_of], pos_in]
        rowptr = np.concatenate([(t["RP"][tau][:, :T] + eoff[:, None]).ravel(), [E]])
        t_rowptr = np.concatenate([(t["TRP"][tau][:, :T] + eoff[:, None]).ravel(), [E]])
        band = t["BAND"][tau].ravel()

        def listed(table, counts, dm):
            if not int(counts[tau].sum()):
                return torch.zeros(0, dtype=torch.int32), 0
            v, so, _ = _ragged(table, counts, tau)
            return torch.from_numpy((v + noff[so]).astype(np.int32)), int(int(dm[tau].max()) <= HEAVY_IN_LAUNCH_DEGREE)
        heavy, mode = listed(t["HV"], t["nh"], t["dmax"])
        t_heavy, t_mode = listed(t["THV"], t["nth"], t["tdmax"])
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32))
        graph = CSRGraph(i32(rowptr), i32(col), i32(t_rowptr), i32(t_col), torch.from_numpy(np.ascontiguousarray(t_wgt)), B * T,
                         heavy, t_heavy, mode, t_mode, torch.from_numpy(np.ascontiguousarray(band)))
        out = (torch.from_numpy(np.ascontiguousarray(ei)), graph)
        if len(self._assembled) < 64:
            self._assembled[key] = out
        return out


class SyntheticResidentDataset(SyntheticTaskDataset):
    """Synthetic counterpart of the reference's frame datasets on top of a device-resident feature store: a few
    synthetic "videos" ([frames, F] arrays, as the reference's ``.npy`` per video) and, per sample, T action windows
    over one of them; ``__getitem__`` does the reference's index arithmetic (feature_store.window_rows: random
    segment sampling on ``train``, uniform otherwise) and returns ``x_idx [T, S]`` instead of ``x``.
    ``host_item`` builds the same sample the reference's way (``np.take`` on the host) from the same random stream."""

    def __init__(self, task: str, length: int, T: int, num_segments: int = 3, features_size: int = 1536,
                 num_class_labels=(115, 478), k: int = 1, seed: int = 1, transform=None, split: str = "train",
                 n_videos: int = 4, frames: int = 600):
        super().__init__(task, length, T, num_segments, features_size, num_class_labels, k, seed, transform)
        import numpy as np
        self.split = split
        rs = np.random.RandomState(seed)
        self.videos = {f"video_{v}": rs.standard_normal((frames, features_size)).astype(np.float32) for v in range(n_videos)}
        self.first_row, off = {}, 0
        for uid, arr in self.videos.items():  # the row layout FeatureStore(self.videos) will use
            self.first_row[uid] = off
            off += arr.shape[0]
        self.windows = []
        for i in range(length):
            uid = f"video_{int(rs.randint(n_videos))}"
            starts = np.sort(rs.randint(0, frames - 8, size=T))
            lens = rs.randint(0, 40, size=T)  # some windows are empty or shorter than S: the zero-clip / linspace paths
            self.windows.append((uid, starts, starts + lens))
        self.rng = np.random.RandomState(seed + 17)

    def _labels(self, i: int):
        return SyntheticTaskDataset.__getitem__(self, i, with_x=False)  # labels / positions / edges of the parent

    def _rows(self, i: int):
        import numpy as np
        from .feature_store import window_rows
        uid, starts, ends = self.windows[i]
        n = self.videos[uid].shape[0]
        rows = [window_rows(self.first_row[uid], n, int(a), int(b), self.S, self.split == "train", self.rng)
                for a, b in zip(starts, ends)]
        return np.stack(rows)

    def __getitem__(self, i: int) -> Data:
        d = self._labels(i)
        d.x_idx = torch.from_numpy(self._rows(i))
        return d

    # -- whole batches (see "whole-batch builders" above) --------------------------------------------------------------------
    def _tables(self):
        """Per-sample labels / positions / scalar attributes / graph templates, built once (they are functions of the seed)."""
        import numpy as np
        tb = getattr(self, "_batch_tables", None)
        if tb is None:
            tmpl = GraphTemplates(self.T)
            ys, poss, tau, scalars = [], [], [], {}
            for i in range(self.length):
                d = self._labels(i)
                ys.append(d.y if torch.is_tensor(d.y) else torch.tensor([d.y]))
                poss.append(d.pos)
                tau.append(tmpl.add(d.edge_index))
                for key, v in vars(d).items():
                    if key not in ("x", "x_idx", "y", "pos", "edge_index", "batch", "edge_attr") and isinstance(v, (int, float)):
                        scalars.setdefault(key, []).append(v)
            vids = list(self.videos)
            vid = np.array([vids.index(w[0]) for w in self.windows], dtype=np.int64)
            tb = self._batch_tables = dict(
                y=torch.stack(ys).numpy(), pos=torch.stack(poss).numpy(), tau=np.array(tau, dtype=np.int64), tmpl=tmpl,
                scalars={k: torch.tensor(v).numpy() for k, v in scalars.items()},
                first=np.array([self.first_row[u] for u in vids], dtype=np.int64)[vid],
                vlen=np.array([self.videos[u].shape[0] for u in vids], dtype=np.int64)[vid],
                starts=np.stack([w[1] for w in self.windows]).astype(np.int64), ends=np.stack([w[2] for w in self.windows]).astype(np.int64))
        return tb

    # -- the same batch from ONE native call (egk_host_build_batch): the loops' default ---------------------------------------------
    # The vectorised builder below costs 0.2-0.3 ms of interpreter time per task batch -- with three live loaders more than half of
    # what the training thread has per 1.3 ms step -- so the entry points ran 10-17 % behind the bench lines.  The per-sample tables are
    # handed to the library once; a batch is then a handful of output allocations + one call that holds no Python object.
    native_batches = True  # False: the numpy builder (the two are tested equal, field for field and in the random stream)

    def _native_tables(self):
        import ctypes as C
        import numpy as np
        from . import _lib
        nt = getattr(self, "_native", None)
        if nt is None:
            tb = self._tables()
            t = tb["tmpl"].tables()
            c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
            y = tb["y"]
            keep = dict(y=c(y.reshape(y.shape[0], -1), np.int64), pos=c(tb["pos"], np.int64), tau=c(tb["tau"], np.int64),
                        first=c(tb["first"], np.int64), vlen=c(tb["vlen"], np.int64), starts=c(tb["starts"], np.int64),
                        ends=c(tb["ends"], np.int64), t_e=c(t["e"], np.int64), t_ei=c(t["EI"], np.int64), t_col=c(t["COL"], np.int64),
                        t_tcol=c(t["TCOL"], np.int64), t_tw=c(t["TW"], np.float32), t_rp=c(t["RP"], np.int64), t_trp=c(t["TRP"], np.int64),
                        t_band=c(t["BAND"], np.uint8), t_nh=c(t["nh"], np.int64), t_nth=c(t["nth"], np.int64), t_hv=c(t["HV"], np.int64),
                        t_thv=c(t["THV"], np.int64), t_dmax=c(t["dmax"], np.int64), t_tdmax=c(t["tdmax"], np.int64))
            heads = int(y.shape[2]) if (y.ndim == 3 and y.shape[1] == self.T) else 0
            d = _lib.HostDataset(T=self.T, S=self.S, train=int(self.split == "train"), y_heads=heads, L=int(y.shape[0]),
                                 y_elems=int(keep["y"].shape[1]), heavy_in_launch=int(HEAVY_IN_LAUNCH_DEGREE),
                                 live_share=float(LIVE_ROWS_MAX_SHARE), n_tmpl=int(keep["t_e"].shape[0]), e_max=int(keep["t_ei"].shape[2]),
                                 h_max=int(keep["t_hv"].shape[1]), th_max=int(keep["t_thv"].shape[1]),
                                 **{k: v.ctypes.data for k, v in keep.items()})
            nt = self._native = dict(desc=d, keep=keep, y_shape=tuple(y.shape[1:]), heads=heads, n_templates=len(tb["tmpl"].items),
                                     hil=int(HEAVY_IN_LAUNCH_DEGREE), share=float(LIVE_ROWS_MAX_SHARE), layouts={},
                                     sizes=np.empty(4, dtype=np.int64))
        return nt

    def batch(self, chunk) -> Data:
        """The batch ``collate([self[i] for i in chunk])`` builds -- field for field, consuming ``self.rng`` exactly as the per-sample
        calls do: by the library's host builder (``native_batches``) straight into ONE buffer (``Arena``: the fields are views of it,
        built when read), else by ``batch_numpy``."""
        if not self.native_batches:
            return self.batch_numpy(chunk)
        import ctypes as C
        import numpy as np
        from . import _lib
        from .feature_store import _mt_commit, _mt_state
        tb = self._tables()
        nt = self._native_tables()
        if (nt["n_templates"] != len(tb["tmpl"].items) or nt["hil"] != int(HEAVY_IN_LAUNCH_DEGREE)
                or nt["share"] != float(LIVE_ROWS_MAX_SHARE)):
            self._native = None  # (the tables or a module-level threshold moved: hand them over again)
            nt = self._native_tables()
        idx = np.ascontiguousarray(list(chunk) if not isinstance(chunk, np.ndarray) else chunk, dtype=np.int64)
        B, T, S, heads = int(idx.shape[0]), self.T, self.S, nt["heads"]
        if B == 0:
            return self.batch_numpy(chunk)
        lib, n = _lib.load(), B * T
        sizes = nt["sizes"]
        rc = int(lib.egk_host_batch_sizes(C.byref(nt["desc"]), idx.ctypes.data, B, sizes.ctypes.data))
        if rc != 0:
            raise ValueError(f"batch: sample index out of range (egk_host_batch_sizes: code {rc})")
        E, nh, nth, n_live = (int(v) for v in sizes)
        live_cap = max(64, (n_live + 63) // 64 * 64) if n_live >= 0 else 0
        key = (B, edge_capacity(E), nh, nth, live_cap)
        lay = nt["layouts"].get(key)
        if lay is None:
            i64, i32 = torch.int64, torch.int32
            ys = nt["y_shape"]
            y_shape = (n, *ys[1:]) if (len(ys) >= 1 and ys != (1,)) else (B * int(np.prod(ys, dtype=np.int64)),)
            fields = [("y", y_shape, i64, False), ("pos", (n,), i64, False), ("edge_index", (2, E), i64, True), ("batch", (n,), i64, False),
                      ("ptr", (B + 1,), i64, False), ("x_idx", (n, S), i64, False), ("graph.rowptr", (n + 1,), i32, False),
                      ("graph.col", (E,), i32, True), ("graph.t_rowptr", (n + 1,), i32, False), ("graph.t_col", (E,), i32, True),
                      ("graph.t_wgt", (E,), torch.float32, True), ("graph.heavy", (nh,), i32, False), ("graph.t_heavy", (nth,), i32, False),
                      ("graph.band", (n,), torch.uint8, False), ("ptr32", (B + 1,), i32, False), ("seg_ptr", (2,), i32, False)]
            if live_cap:
                fields += [("live_idx", (live_cap,), i64, False), ("live_inv", (n,), i64, False), ("live_y", (live_cap, heads), i64, False)]
            fields += [(f"attr.{k}", (B,), torch.from_numpy(v[:1]).dtype, False) for k, v in tb["scalars"].items()]
            lay = nt["layouts"][key] = ArenaLayout("task", fields, edge_capacity(E))
        buf = np.empty(lay.total, dtype=np.uint8)
        base, offs = buf.ctypes.data, lay.offs
        ob = _lib.HostBatch(E=E, heavy_cap=nh, t_heavy_cap=nth, live_cap=live_cap, edge_cap=lay.edge_cap,
                            **{fld: base + offs[name] for name, fld in (_BATCH_PTRS_LIVE if live_cap else _BATCH_PTRS)})
        if self.split == "train":
            state, mkey, mpos = _mt_state(self.rng)
            rc = int(lib.egk_host_build_batch(C.byref(nt["desc"]), mkey.ctypes.data, mpos.ctypes.data, idx.ctypes.data, B, C.byref(ob)))
            if rc == 0:
                _mt_commit(self.rng, state, mkey, mpos)
        else:
            rc = int(lib.egk_host_build_batch(C.byref(nt["desc"]), None, None, idx.ctypes.data, B, C.byref(ob)))
        if rc != 0 or int(ob.n_live) != n_live:
            raise ValueError(f"egk_host_build_batch failed (code {rc})")
        seg = buf[offs["seg_ptr"]:offs["seg_ptr"] + 8].view(np.int32)
        seg[0], seg[1] = 0, n
        for k, v in tb["scalars"].items():
            o = offs[f"attr.{k}"]
            buf[o:o + B * v.itemsize].view(v.dtype)[:] = v[idx]
        meta = dict(num_graphs=B, num_nodes=n, n_heavy=nh, n_t_heavy=nth, heavy_mode=int(ob.heavy_mode), t_heavy_mode=int(ob.t_heavy_mode),
                    pos_range=(int(ob.pos_min), int(ob.pos_max)))
        if live_cap and ob.live_ap_step > 0:
            meta["live_ap"] = (int(ob.live_ap_first), int(ob.live_ap_step), live_cap)
        out = arena_batch(Arena(buf, lay, E, meta))
        # graph-structure fingerprint (engine.structure_key): the samples' graph templates determine the edges, B and T the sequences
        out.__dict__["_struct_key"] = hash((id(self), B, nt["keep"]["tau"][idx].tobytes())) or 1
        return out

    def batch_numpy(self, chunk) -> Data:
        """``collate([self[i] for i in chunk])`` -- field for field, and consuming ``self.rng`` exactly as the per-sample calls
        do -- assembled with vector arithmetic."""
        import numpy as np
        from .feature_store import window_rows_batch
        tb = self._tables()
        idx = np.asarray(list(chunk), dtype=np.int64)
        B, T = idx.shape[0], self.T
        rows = window_rows_batch(np.repeat(tb["first"][idx], T), np.repeat(tb["vlen"][idx], T), tb["starts"][idx].ravel(),
                                 tb["ends"][idx].ravel(), self.S, self.split == "train", self.rng)
        # (numpy throughout, one zero-copy torch view per array at the end: a torch CPU op on a tiny tensor costs more than the
        #  index arithmetic of the whole batch when the intra-op pool has many threads)
        tn = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        y = tb["y"][idx]
        y = y.reshape(-1, *y.shape[2:]) if y.ndim > 1 and y.shape[1:] != (1,) else y.reshape(-1)
        pos = tb["pos"][idx].reshape(-1)
        ei, graph = tb["tmpl"].assemble(tb["tau"][idx])
        n = B * T
        ptr = np.arange(0, n + 1, T, dtype=np.int64)
        out = Data(x=None, y=tn(y), pos=tn(pos), edge_index=ei, batch=tn(np.repeat(np.arange(B, dtype=np.int64), T)), ptr=tn(ptr),
                   num_graphs=B)
        if n:
            out.pos_range = (int(pos.min()), int(pos.max()))
        out.x_idx = torch.from_numpy(rows)
        out.graph = graph
        out.ptr32 = tn(ptr.astype(np.int32))
        out.seg_ptr = tn(np.array([0, n], dtype=np.int32))
        _attach_live_rows(out)
        for key, v in tb["scalars"].items():
            setattr(out, key, tn(v[idx]))
        return out

    def host_item(self, i: int) -> Data:
        """The sample as the reference builds it: features taken on the host (consumes the same random numbers)."""
        import numpy as np
        d = self._labels(i)
        rows = self._rows(i)
        table = np.concatenate(list(self.videos.values()))
        d.x = torch.from_numpy(np.where(rows[..., None] >= 0, table[np.maximum(rows, 0)], 0.0).astype(np.float32))
        return d
