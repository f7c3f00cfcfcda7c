"""Training steps of the hot path (reference main_temporal.train :49-134, main_egopack.train
:64-159 and train_step_task :45-61) re-designed for MI355X:

  * the backbone runs ONCE per step over the node sets of all enabled task batches (fused pass,
    per-task graph-LayerNorm statistics), then each task's head / loss runs on its row slice;
  * the objective ``sum_t weight_t * loss_t.mean()`` is one deterministic reduction chain;
  * gradients accumulate inside the backward kernels into the optimiser's flat buffer, are
    exchanged by dist.GradSync (RCCL) when world_size > 1, and Adam is one launch;
  * ``capture()`` records forward + backward (+ Adam when single-GPU) of a fixed-shape step into a
    hipGraph (torch.cuda.CUDAGraph) so a replayed step costs one graph launch instead of ~300
    kernel launches from Python.
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Dict, Mapping, Optional, Sequence

import torch

from . import ops, switches
from .data import Data, merge_batches
from .criterion import MetricSelectorWrapper
from .dist import GradSync

TASK_ORDER = ("ar", "lta", "oscc", "pnr")  # order of the loss terms in main_temporal.train


STRUCTURE_FIELDS = ("edge_index", "batch", "ptr", "ptr32", "graph")


def structure_key(b: Data) -> int:
    """Host-side fingerprint of a batch's graph structure: sequence boundaries and edges (the CSR arrays derive from them)."""
    if b.__dict__.get("_arena") is not None and b.__dict__.get("_struct_key"):
        return b.__dict__["_struct_key"]  # (the native builder's: a hash of the samples' graph templates)
    parts = []
    for name in ("edge_index", "ptr"):
        v = getattr(b, name, None)
        if torch.is_tensor(v) and v.device.type == "cpu":
            parts.append((name, tuple(v.shape), hash(v.contiguous().numpy().tobytes())))
        elif torch.is_tensor(v):
            return 0  # (already on a device: no cheap fingerprint -- 0 never matches, everything is copied)
    return hash(tuple(parts)) or 1


def stage_batches(host, device, order=TASK_ORDER, pin: bool = True, store=None, dtype=None, features: bool = True):
    """Host -> device transfer of one step's task batches for the fused pass: the feature blocks are packed
    into ONE (pinned) buffer and moved with ONE copy; the returned per-task batches view row ranges of the
    device buffer and ``merged`` exposes the whole buffer (first contraction at M = all nodes).

    With a ``feature_store.FeatureStore`` and batches that carry ``x_idx`` instead of ``x`` nothing but the index
    matrices crosses PCIe: ONE gather launch builds the packed ``[sum N, S, F]`` buffer in HBM."""
    live = [t for t in order if host.get(t) is not None]
    resident = store is not None and all(getattr(host[t], "x", None) is None and getattr(host[t], "x_idx", None) is not None
                                         for t in live)
    if resident and all(host[t].__dict__.get("_arena") is not None for t in live):
        return _stage_arenas([host[t] for t in live], live, device, store, dtype, features)
    if resident:
        idx = torch.cat([host[t].x_idx for t in live])
        if pin and idx.device.type == "cpu":
            idx = idx.pin_memory()
        dbuf = store.gather(idx.to(device, non_blocking=True), dtype=dtype)
        rows = [host[t].x_idx.shape[0] for t in live]
    else:
        from .data import pack_features
        slot = []
        buf = pack_features([host[t] for t in live], pin=pin, ring_slot=slot)
        dbuf = buf.to(device, non_blocking=True)
        if slot and dbuf.is_cuda:  # the pinned staging buffer may be handed out again once this copy has completed
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            slot[0][1] = ev
        rows = [host[t].x.shape[0] for t in live]
    from .data import to_device_packed
    dev, off = {}, 0
    saved = {}
    for t in live:  # everything but the features (they are already on the device) travels in ONE packed copy
        b = host[t]
        saved[t] = (b.x, getattr(b, "x_base", None), getattr(b, "x_idx", None))
        b.x = b.x_base = b.x_idx = None
        b._struct_key = structure_key(b)  # (lets merge_batches reuse the merged CSR of equal structures)
    placeholder = [torch.empty(0)] * len(live)
    for t, ph in zip(live, placeholder):  # merge on the host; the packed device buffer becomes its x below
        host[t].x, host[t].x_base = ph, None
    merged = merge_batches([host[t] for t in live])
    merged.x = None
    for t in live:
        host[t].x = host[t].x_base = None
    moved = to_device_packed([*(host[t] for t in live), merged], device)
    for t, n, d in zip(live, rows, moved[:-1]):
        d.x = dbuf[off:off + n]
        d.x_base = dbuf
        off += n
        dev[t] = d
    for t in live:
        host[t].x, host[t].x_base, host[t].x_idx = saved[t]
    md = moved[-1]
    md.x = md.x_base = dbuf
    if getattr(md, "_blob", None) is not None:
        md._blob.names = [*live, "merged"]
    # fingerprint of everything that describes the graphs (not the features / labels): equal keys = equal structure, so a
    # replay on static buffers (StepBase.train_step) need not rewrite the CSR arrays, edge lists and batch vectors
    for t in live:
        dev[t]._struct_key = host[t]._struct_key
    md._struct_key = hash(tuple(dev[t]._struct_key for t in live)) or 1
    return dev, md


def _stage_arenas(hs, live, device, store, dtype, features: bool = True):
    """``stage_batches`` for batches the native builder wrote into arenas (data.Arena): the merged batch comes from one host call,
    every batch travels as one memcpy into the transfer buffer -- the store rows with it -- and the packed feature block is gathered
    task by task from the device copy of those rows.  No tensor of the step is touched on the host.  ``features=False``: the block
    is NOT gathered here -- the transfer carries the store (``BlobRef.store``) and whoever consumes the batches gathers the rows
    where it wants them (a captured training step: straight into its idle input buffers, StepBase._input_slot;
    ``ensure_features`` for anything else)."""
    from .data import to_device_packed
    merged = merge_batches(hs)
    merged.x = None
    moved = to_device_packed([*hs, merged], device)
    rows = [h._arena.meta["num_nodes"] for h in hs]
    idx0 = moved[0].__dict__["x_idx"]
    x_dtype = dtype or store.table.dtype
    dbuf = torch.empty((sum(rows), idx0.shape[1], store.features_size), dtype=x_dtype, device=idx0.device) if features else None
    dev, off = {}, 0
    for t, n, h, d in zip(live, rows, hs, moved[:-1]):
        if features:
            store.gather(d.__dict__["x_idx"], out=dbuf[off:off + n], dtype=dtype)
            d.x = dbuf[off:off + n]
        d.x_base = dbuf
        d._struct_key = h.__dict__.get("_struct_key", 0)
        dev[t] = d
        off += n
    md = moved[-1]
    md.x = md.x_base = dbuf
    ref = getattr(md, "_blob", None)
    if ref is not None:
        ref.names = [*live, "merged"]
        ref.store, ref.x_dtype = store, x_dtype  # (what a consumer that gathers the rows itself needs)
    md._struct_key = hash(tuple(dev[t]._struct_key for t in live)) or 1
    return dev, md


def ensure_features(batches, merged=None) -> None:
    """Give batches that were staged without their feature block (``stage_batches(features=False)``) the block: gathered from the
    transfer's store on the current stream (no-op for batches that have it)."""
    live = [(t, b) for t, b in batches.items() if b is not None]
    if not live or all(b.__dict__.get("x") is not None for _, b in live):
        return
    ref = _shared_blob(batches, merged)
    store = getattr(ref, "store", None) if ref is not None else None
    if store is None:
        raise RuntimeError("batches without features and without a feature store to gather them from")
    idx = [b.__dict__["x_idx"] for _, b in live]
    dbuf = torch.empty((sum(i.shape[0] for i in idx), idx[0].shape[1], store.features_size), dtype=ref.x_dtype, device=idx[0].device)
    off = 0
    for (t, b), i in zip(live, idx):
        n = i.shape[0]
        store.gather(i, out=dbuf[off:off + n], dtype=ref.x_dtype)
        b.x, b.x_base = dbuf[off:off + n], (dbuf if merged is not None else None)
        off += n
    if merged is not None:
        merged.x = merged.x_base = dbuf


_COPY_STREAMS = {}  # device index -> the staging copy stream (StagedBatches)


STAGE_LOCK = threading.RLock()  # held by StagedBatches' thread while it stages, and by eager steps / captures (see ``_ahead``)


class StagedBatches:
    """Iterate a host-batch iterator one step AHEAD on a copy stream: while the device runs step i, the host collates batch
    i + 1 and its tensors cross PCIe (the packed feature block is 57 MB per step of the headline workload: 1.0 ms at 56
    GB/s, which would otherwise sit between two steps on the compute stream).  Yields ``(batches, merged)`` as
    ``stage_batches`` returns them (``merged`` None for a single task / per-task backbone passes), already ordered behind
    the copy on the consumer's stream."""

    def __init__(self, host_iter, device, order=TASK_ORDER, fused: bool = True, store=None, dtype=None, depth: Optional[int] = None,
                 step=None):
        self.it, self.device, self.order, self.fused, self.store, self.dtype = iter(host_iter), device, order, fused, store, dtype
        # the training step these batches feed (optional): once it gathers the store rows into its own input buffers
        # (``StepBase.gathers_inputs``) the feature block is no longer gathered here
        self.step = step
        # steps staged ahead by the staging thread (0: staged inline, right behind the consumer's launch of the step before)
        self.depth = (2 if switches.enabled("staging_thread") else 0) if depth is None else int(depth)
        # ONE copy stream per device for the life of the process: the caching allocator keeps a pool per stream, so a fresh
        # stream per epoch stranded every epoch's staging blocks in a pool nobody allocates from again (reserved memory grew
        # by ~130 MB per epoch of the headline workload while the allocated bytes stayed flat)
        self.copy_stream = None
        if torch.device(device).type == "cuda":
            key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
            self.copy_stream = _COPY_STREAMS.get(key)
            if self.copy_stream is None:
                # (a priority stream: the runtime keeps hardware queues per priority level, so the staging launches do not sit in a
                #  queue behind a branch of the replayed graph that is waiting for its dependencies)
                prio = int(switches.value("staging_priority", -1)) if switches.enabled("staging_priority") else 0
                self.copy_stream = _COPY_STREAMS[key] = torch.cuda.Stream(device=device, priority=prio)

    def _stage(self, host):
        live = {t: b for t, b in host.items() if b is not None}
        features = not (self.store is not None and getattr(self.step, "gathers_inputs", False))
        if self.fused and len(live) > 1:
            return stage_batches(live, self.device, self.order, store=self.store, dtype=self.dtype, features=features)
        from .data import to_device_packed
        moved = to_device_packed([live[t] for t in live], self.device)  # (one copy for all of a step's tensors)
        ref = getattr(moved[0], "_blob", None) if moved else None
        if ref is not None:
            ref.names = list(live)
            if self.store is not None:
                ref.store, ref.x_dtype = self.store, self.dtype or self.store.table.dtype
        for t, d in zip(live, moved):
            d._struct_key = structure_key(live[t])
            if self.store is not None and getattr(d, "x", None) is None and getattr(d, "x_idx", None) is not None:
                if features or ref is None or "x_idx" not in d.__dict__:
                    d.x = self.store.gather(d.x_idx, dtype=self.dtype)  # features from the device-resident table
        return dict(zip(live, moved)), None

    def _fetch(self):
        try:
            host = next(self.it)
        except StopIteration:
            return None
        if self.copy_stream is None:
            return self._stage(host), None
        # (no wait for the compute stream: the caching allocator keeps per-stream pools, and what is allocated here is
        #  marked as used by the consumer's stream below)
        with torch.cuda.stream(self.copy_stream):
            staged = self._stage(host)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        return staged, done

    def _ahead(self):
        """``_fetch`` results from a thread of their own, ``depth`` steps ahead (None ends the walk, an exception is re-raised in
        the consumer).  The thread holds ``STAGE_LOCK`` while it stages: eager steps and captures take the same lock, so that
        nothing of another thread touches the HIP runtime while a capture is open and the launch-order bookkeeping of ``ops``
        has ONE writer; replays of the captured step -- every steady-state step -- do not take it and overlap the staging."""
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                torch.cuda.set_device(self.copy_stream.device)
                while not stop.is_set():
                    with STAGE_LOCK:
                        nxt = self._fetch()
                    if not put(nxt) or nxt is None:
                        return
            except BaseException as e:  # noqa: BLE001 (handed to the consumer)
                put(e)
        if switches.debug("stage_profile"):
            inner = produce

            def produce():  # noqa: F811 (development: where the staging thread's time goes)
                import cProfile
                import pstats
                pr = cProfile.Profile()
                try:
                    pr.runcall(inner)
                finally:
                    pstats.Stats(pr).sort_stats("tottime").print_stats(35)
        th = threading.Thread(target=produce, name="egk-staging", daemon=True)
        th.start()
        try:
            while True:
                nxt = q.get()
                if nxt is None:
                    return
                if isinstance(nxt, BaseException):
                    raise nxt
                yield nxt
        finally:
            stop.set()
            th.join(timeout=10.0)

    def __iter__(self):
        def inline():
            nxt = self._fetch()
            while nxt is not None:
                yield nxt
                nxt = self._fetch()  # issued right after the consumer launched its step: overlaps it
        for nxt in (self._ahead() if (self.copy_stream is not None and self.depth > 0) else inline()):
            (batches, merged), done = nxt
            if done is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(done)
                ref = _shared_blob(batches, merged)
                if ref is not None:  # every tensor but the features is a view of the transfer's one buffer
                    ref.ready = done  # (what a consumer on another stream waits for: StepBase._input_slot)
                    used = [ref.dev, *(getattr(b, k, None) for b in [*batches.values(), merged] if b is not None
                                       for k in ("x", "x_base"))]
                else:
                    used = [v for _, v in [*(kv for b in batches.values() for kv in _walk(b)), *_walk(merged)]]
                for v in used:
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)  # allocated on the copy stream, consumed here
            yield batches, merged


# ---- static-shape batches: what a captured step may be replayed on ----------------------------------------------------
def _walk(obj, prefix=""):
    """(path, value) for every tensor / plain scalar reachable from a Data / CSRGraph (lists of tensors included)."""
    from dataclasses import fields, is_dataclass
    if obj is None:
        return
    if torch.is_tensor(obj):
        yield prefix, obj
    elif is_dataclass(obj):
        for f in fields(obj):
            yield from _walk(getattr(obj, f.name), f"{prefix}.{f.name}")
    elif isinstance(obj, Data):
        obj.materialise()
        for k in sorted(obj.__dict__):
            yield from _walk(obj.__dict__[k], f"{prefix}.{k}")
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _walk(v, f"{prefix}[{i}]")
    elif isinstance(obj, (int, float, bool, str)):
        yield prefix, obj


# Edge-sized arrays live in the static buffers of a captured step at a CAPACITY (data.EDGE_FIELDS / EDGE_BUCKET): the LTA edge
# count moves by a few entries from batch to batch, no kernel takes E, and the signature compares capacities.
from .data import EDGE_BUCKET, EDGE_FIELDS, edge_capacity as _edge_capacity  # noqa: E402


def _is_edge_field(path: str) -> bool:
    return path.endswith(EDGE_FIELDS)


def batch_signature(batches: Mapping[str, Data], merged=None) -> tuple:
    """Everything a capture bakes in: the shape / dtype of every tensor (edge-sized arrays: their capacity) and every plain
    scalar (node counts, heavy-row modes ...) of the step's batches.  Two steps with equal signatures differ in tensor
    VALUES only.  Private attributes (``_struct_key`` ...) describe values, not shapes: they are not part of it."""
    sig = []
    for name, obj in [*sorted(batches.items()), ("merged", merged)]:
        for path, v in _walk(obj, name):
            if "._" in path or (torch.is_tensor(v) and v.dim() >= 1 and path.endswith(".x_base")):  # (x_base aliases x)
                continue
            if torch.is_tensor(v):
                shape = tuple(v.shape)
                if _is_edge_field(path) and v.dim() >= 1:
                    shape = shape[:-1] + (_edge_capacity(shape[-1]),)
                sig.append((path, (shape, v.dtype)))
            else:
                sig.append((path, v))
    return tuple(sig)


def _fast_key(batches, merged):
    """A cheap stand-in for ``batch_signature`` when every batch carries a structure fingerprint: the fingerprints and the
    feature blocks' shapes / types (None: some batch has no fingerprint -- take the full signature)."""
    parts = []
    for name, b in [*sorted(batches.items()), ("merged", merged)]:
        if b is None:
            parts.append((name, None))
            continue
        k = getattr(b, "_struct_key", 0)
        x = getattr(b, "x", None)
        if not k or not torch.is_tensor(x):
            return None
        y = getattr(b, "y", None)
        # the labelled-row set of a compacted head is baked into a capture too (the strided row view ``live_ap`` and the padded
        # height of ``live_idx`` / ``live_y``): it depends on label VALUES, which the structure fingerprint does not cover
        li = getattr(b, "live_idx", None)
        parts.append((name, k, tuple(x.shape), x.dtype, tuple(y.shape) if torch.is_tensor(y) else None,
                      getattr(b, "live_ap", None), tuple(li.shape) if torch.is_tensor(li) else None))
    return tuple(parts)


def _shared_blob(batches, merged):
    """The ``data.BlobRef`` of a step whose batches all came out of ONE packed transfer with a layout signature, else None."""
    objs = [b for b in [*batches.values(), merged] if b is not None]
    ref = getattr(objs[0], "_blob", None) if objs else None
    if ref is None or ref.gsig is None or ref.names is None or any(getattr(b, "_blob", None) is not ref for b in objs):
        return None
    return ref


def _feature_key(batches, merged):
    """(name, shape, dtype) of every feature block of a step's batches -- also for batches staged without their block
    (``stage_batches(features=False)``): its shape follows from the store rows, its type is the transfer's."""
    ref = None
    out = []
    for name, b in [*sorted(batches.items()), ("merged", merged)]:
        if b is None:
            continue
        x = b.__dict__.get("x", None) if "x" in b.__dict__ else getattr(b, "x", None)
        if torch.is_tensor(x):
            out.append((name, tuple(x.shape), x.dtype))
            continue
        ref = ref or _shared_blob(batches, merged)
        store = getattr(ref, "store", None) if ref is not None else None
        if store is None:
            continue
        if name == "merged":
            idx = [bb.__dict__.get("x_idx") for _, bb in sorted(batches.items()) if bb is not None]
            if all(torch.is_tensor(i) for i in idx):
                out.append((name, (sum(i.shape[0] for i in idx), idx[0].shape[1], store.features_size), ref.x_dtype))
        else:
            i = b.__dict__.get("x_idx")
            if torch.is_tensor(i):
                out.append((name, (*i.shape, store.features_size), ref.x_dtype))
    return tuple(out)


def _static_from_blob(ref, batches, merged):
    """Private buffers for a capture out of a packed transfer: ONE byte buffer of the transfer's layout (edge-sized arrays at
    capacity) + clones of the feature blocks.  -> (buffer, static batches, static merged)."""
    blob = torch.empty_like(ref.dev)
    blob.copy_(ref.dev)
    outs = dict(zip(ref.names, ref.rebuild(blob, True)))
    static_m = outs.pop("merged", None) if merged is not None else None
    static_b = {t: outs.get(t) for t in batches}
    for dst, src in [*((static_b[t], batches[t]) for t in batches if batches[t] is not None), (static_m, merged)]:
        if dst is None:
            continue
        for k, v in src.__dict__.items():  # what was attached after the transfer (features, fingerprints)
            if k in ("_blob", "_fill") or k in dst.__dict__ and dst.__dict__[k] is not None:
                continue
            setattr(dst, k, v)
    if static_m is not None and torch.is_tensor(merged.x):
        static_m.x = merged.x.clone()
        _rewire_packed(static_b, static_m)
    else:
        for t, b in static_b.items():
            if b is not None and torch.is_tensor(b.x):
                b.x = batches[t].x.clone()
                b.x_base = None
    return blob, static_b, static_m


@torch.no_grad()
def copy_batch_values(dst_batches, dst_merged, src_batches, src_merged, dst_walks=None) -> None:
    """dst <- src for every tensor of two sets of batches with the same signature (device to device).  Per-task feature
    blocks that are row ranges of a packed buffer (``x_base``) are written once, through the buffer; edge-sized arrays are
    written into the first E entries of their capacity.  Graph-structure arrays are skipped when both sides carry the same
    non-zero structure fingerprint; after a full copy the destination takes over the source's fingerprint."""
    done = set()
    pairs = [(dst_batches[t], src_batches[t]) for t in sorted(dst_batches) if dst_batches[t] is not None]
    if dst_merged is not None:
        pairs.append((dst_merged, src_merged))
    for dst, src in pairs:
        smap = dict(_walk(src))
        base = getattr(dst, "x_base", None)
        k_dst, k_src = getattr(dst, "_struct_key", 0), getattr(src, "_struct_key", 0)
        same_structure = bool(k_dst) and k_dst == k_src
        if dst_walks is not None:  # (the destination's tensors never change: walked once per static batch)
            walk = dst_walks.get(id(dst))
            if walk is None:
                walk = dst_walks[id(dst)] = [(p_, d_) for p_, d_ in _walk(dst) if torch.is_tensor(d_) and not p_.endswith(".x_base")]
        else:
            walk = _walk(dst)
        for path, d in walk:
            if not torch.is_tensor(d) or path.endswith(".x_base"):
                continue
            if same_structure and path.split(".")[1].split("[")[0] in STRUCTURE_FIELDS:
                continue  # equal fingerprints (stage_batches): the graph arrays already hold these values
            s_ = smap.get(path)
            if s_ is None or d.data_ptr() == s_.data_ptr():
                continue
            if (path == ".x" and torch.is_tensor(base) and base.untyped_storage().data_ptr() == d.untyped_storage().data_ptr()
                    and d.numel() != base.numel()):
                continue  # a row range of the packed buffer: the buffer itself is copied as the merged batch's x
            key = (d.untyped_storage().data_ptr(), d.storage_offset(), tuple(d.shape))
            if key not in done:
                if d.shape != s_.shape and _is_edge_field(path):
                    d[..., :s_.shape[-1]].copy_(s_, non_blocking=True)  # (the tail keeps stale entries no row range reaches)
                else:
                    d.copy_(s_, non_blocking=True)
                done.add(key)
        if not same_structure:
            dst._struct_key = k_src  # (0 when the source's structure is unknown: the next copy is a full one again)


def _clone_batch(d: Data, pad_edges: bool = False) -> Data:
    """Deep copy of a device batch (own storage for every tensor; scalars shared).  ``pad_edges``: edge-sized arrays are
    allocated at their capacity (zeros behind the copied entries) -- the static buffers of a captured training step."""
    from dataclasses import fields, is_dataclass, replace

    def dup(path, t):
        if pad_edges and _is_edge_field(path) and t.dim() >= 1 and _edge_capacity(t.shape[-1]) != t.shape[-1]:
            out = torch.zeros(t.shape[:-1] + (_edge_capacity(t.shape[-1]),), dtype=t.dtype, device=t.device)
            out[..., :t.shape[-1]].copy_(t)
            return out
        return t.clone()
    out = Data()
    for k, v in d.materialise().__dict__.items():
        if torch.is_tensor(v):
            v = dup(f".{k}", v)
        elif is_dataclass(v):
            v = replace(v, **{f.name: (dup(f".{k}.{f.name}", getattr(v, f.name)) if torch.is_tensor(getattr(v, f.name))
                                       else getattr(v, f.name)) for f in fields(v)})
        elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
            v = [t.clone() for t in v]
        setattr(out, k, v)
    return out


def _rewire_packed(static_b, static_m) -> None:
    """After cloning: the per-task feature blocks become row ranges of the merged batch's packed buffer again."""
    if static_m is None or not torch.is_tensor(getattr(static_m, "x", None)):
        return
    off = 0
    for t in [t for t in TASK_ORDER if static_b.get(t) is not None]:
        b = static_b[t]
        n = b.x.shape[0]
        b.x = static_m.x[off:off + n]
        b.x_base = static_m.x
        off += n
    static_m.x_base = static_m.x


# hipGraph capture mode.  'thread_local': HIP calls of OTHER threads stay legal while this thread captures -- the
# process-group watchdog of torch.distributed polls its events (hipEventQuery) from its own thread at any time, and
# under the default 'global' mode that query is an error that aborts the process.
CAPTURE_MODE = "thread_local"


class StepBase:
    """Shared machinery of the two training steps: fused multi-task backbone pass, eager step with gradient
    exchange + flat Adam, hipGraph capture / replay.  Subclasses implement ``losses(batches, merged)`` returning
    ``(total, vectors, extra)``."""

    order: Sequence[str] = TASK_ORDER
    wgrad_grouping_default = True
    wgrad_group_count = None  # bf16 weight-gradient problems per grouped launch (None: _wgrad_count's policy)

    def _wgrad_count(self, batches, merged=None):
        """H x H weight-gradient problems per grouped launch.  Eight problems are 512 tiles = two whole rounds of the chip (16.4 us per
        problem at 6144 rows against 20 for six = 1.5 rounds), but such a launch holds EVERY CU for both rounds and the backward chain's
        launches wait behind it: worth it while the K walk is short.  Measured, 6 against 8 on one box, alternating
        (tools/round6/wgrad_count_ab*.sh): fused three-task step at 6144 rows 1.308 -> 1.287 ms (its 8-rank dry run 1.360 -> 1.317,
        sharded 1.282 -> 1.247), four tasks at 8192 rows 1.636 -> 1.644, at 16384 rows 2.945 -> 3.013, one task at 2048 rows
        0.837 -> 0.847 -- so: eight for a fused multi-task pass of fewer than 8192 rows, else the module default (six)."""
        if self.wgrad_group_count is not None:
            return self.wgrad_group_count
        live = [b for b in batches.values() if b is not None]
        if self.fused and merged is not None and len(live) > 1:
            rows = getattr(merged, "pos", None)
            if rows is not None and int(rows.shape[0]) < 8192:
                return 8
        return None
    # forked launches (weight gradients, early Adam) issued one launch late so that the dX chain keeps its hardware queue under
    # capture (ops.defer_after_next_launch): -4 % on the multi-task steps, -1.6 % on the single-task step; the EgoPack step,
    # whose GraphONE chains already occupy three queues, measured 4.04 vs 3.51 ms with it and leaves it off
    deferred_forks_default = True

    def _init_base(self, model, tasks, weights, optimizer, fused_backbone, sync, parallel_heads):
        self.model, self.tasks = model, dict(tasks)
        self.weights = {t: float(w) for t, w in weights.items()}
        self.optimizer, self.fused, self.sync = optimizer, fused_backbone, sync
        self.enabled = [t for t in self.order if self.weights.get(t, 0) > 0 and t in self.tasks]
        self.parallel_heads = parallel_heads
        self._head_streams = []
        self._graph = None
        self._static_out = None
        self._static_in = None
        self._fuse_adam = True
        self._stage_state, self._cuts = None, []
        # optional launch(es) in front of every step, inside the captured graph too: e.g. the feature-store gather that
        # materialises the step's input block from its index matrix (feature_store.FeatureStore.gather(idx, out=buffer))
        self.input_hook = None
        # weight-gradient launches of the backbone on a side stream (they feed nothing but the optimizer): 2-4 % on the
        # multi-task steps, 1.8 % on the single-task step (1.236 -> 1.214 ms), neutral on the EgoPack step
        self.wgrad_side_streams = bool(parallel_heads)
        # H x H weight gradients parked and issued four at a time as ONE grouped launch of 256 workgroups (ops._wgrad_defer):
        # -4 % on the 3-task step, -12 % on the single-task step; the EgoPack step measured 3.59 vs 3.46-3.52 ms with it in round 1
        # and left it off until round 4 (EgoPackStep.wgrad_grouping_default)
        self.wgrad_grouping = type(self).wgrad_grouping_default
        self.deferred_forks = type(self).deferred_forks_default
        # development switches (egopack_amd/switches.py): the environment overrides the step class's defaults
        for name in ("wgrad_grouping", "deferred_forks"):
            forced = switches.override(name)
            if forced is not None:
                setattr(self, name, forced)
        if not switches.enabled("grouped_heads"):
            self.grouped_heads = False
        self._fused_loss = switches.enabled("fused_loss")
        if not switches.enabled("ln_fusion"):
            ops._ln_fusion["on"] = False

    # ---- backbone ------------------------------------------------------------------------------------
    def features(self, batches: Mapping[str, Data], merged: Optional[Data] = None) -> Dict[str, torch.Tensor]:
        live = [t for t in self.enabled if batches.get(t) is not None]
        if self.fused and len(live) > 1:
            if merged is None:
                merged = merge_batches([batches[t] for t in live])
                merged = merged.to(batches[live[0]].pos.device)
            feat = self.model(merged)
            parts = ops.split_rows(feat, [batches[t].pos.shape[0] for t in live])
            return dict(zip(live, parts))
        return {t: self.model(batches[t]) for t in live}

    def _run_heads(self, feats, head_fn, main_job=None):
        """``head_fn(t, feat) -> (loss_vector, extra)`` for every task; on side streams when there are several
        (independent half-chip contractions: they overlap).  A head_fn that also runs the head's BACKWARD (see
        ``_backward_pass``) must do so here, inside the head's stream context: one ``backward()`` over several streams
        replays the branches one after the other (tools/exp/branch_overlap.py: 676 us against 441 us for three chains of
        20 contractions), one call per branch inside its stream context lets them overlap."""
        vectors, extras = {}, {}
        if main_job is not None and not feats:
            main_job()
            return vectors, extras
        if self.parallel_heads and (len(feats) > 1 or main_job is not None) and next(iter(feats.values())).is_cuda:
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            n_streams = min(len(feats), getattr(self, "max_head_streams", len(feats)))
            while len(self._head_streams) < n_streams:
                self._head_streams.append(torch.cuda.Stream())
                ops.exclude_wgrad_streams(self._head_streams[-1:])
            used = self._head_streams[:n_streams]
            for st in used:
                st.wait_event(fork)
            if main_job is not None:  # a chain that stays on the main stream, beside the forked ones; issued FIRST: under
                main_job()            # capture the child of a fork that is created first keeps the parent's hardware queue
            for i, (t, feat) in enumerate(feats.items()):
                st = used[i % n_streams]
                feat.record_stream(st)
                with torch.cuda.stream(st):
                    vectors[t], extras[t] = head_fn(t, feat)
            for st in used:
                main.wait_stream(st)
        else:
            for t, feat in feats.items():
                vectors[t], extras[t] = head_fn(t, feat)
            if main_job is not None:
                main_job()
        return vectors, extras

    def _adam_keeps_lo(self) -> bool:
        opt = self.optimizer
        return bool(getattr(opt, "adam_writes_lo", False) and getattr(opt, "flat_w16lo", None) is not None)

    def _scope_streams(self) -> None:
        """This step's own head / task / side streams are the excluded ones from here on (ops.scope_excluded_streams)."""
        g1 = getattr(self, "graphone", None)
        ops.scope_excluded_streams([*getattr(self, "_head_streams", ()), *(getattr(g1, "_task_streams", ()) if g1 is not None else ()),
                                    getattr(self, "_precise_side", None)])

    # ---- running per-task loss sums (what a training loop logs per epoch: reference main_temporal.py:129-134) -----------------------
    # The reference keeps them with ``.item()`` per step; kept as ``sum()`` + ``add_`` launches between two replays of the captured
    # step they were nine launches (48 us) on the stream every replay waits for.  They accumulate INSIDE the step instead: one f64
    # slot per enabled task, added to by the objective's launch (or a one-workgroup launch of their own) beside the parked weight
    # gradients; the host only counts elements.  ``loss_sums()`` reads (and clears) them: one synchronisation per epoch.
    def _loss_acc_for(self, device):
        acc = getattr(self, "_loss_acc", None)
        if acc is None or acc.device != device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the loss accumulators must exist before a capture (run one eager step, or loss_sums(), first)")
            acc = self._loss_acc = torch.zeros(max(len(self.enabled), 1), dtype=torch.float64, device=device)
            self._loss_scratch = torch.zeros(1, dtype=torch.float32, device=device)
        return acc

    @torch.no_grad()
    def _ride_loss_sums(self, vectors) -> None:
        live = [v for v in vectors.values() if v is not None]
        if not live or not live[0].is_cuda:
            return
        acc, order = self._loss_acc_for(live[0].device), list(self.enabled)
        src = {t: v.detach() for t, v in vectors.items()}
        ops.park_rider(lambda: ops.weighted_mean_sum_into(self._loss_scratch, [src.get(t) for t in order], [0.0] * len(order), acc=acc),
                       tuple(src.values()))

    def _count_losses(self, vectors) -> None:
        cnt = self.__dict__.setdefault("_loss_counts", {})
        for t, v in vectors.items():
            cnt[t] = cnt.get(t, 0) + int(v.numel())

    def loss_sums(self, reset: bool = True) -> Dict[str, tuple]:
        """{task: (sum of its per-element losses, elements)} over the steps since the last reset (one device synchronisation)."""
        acc = getattr(self, "_loss_acc", None)
        cnt = self.__dict__.setdefault("_loss_counts", {})
        vals = acc.tolist() if acc is not None else [0.0] * len(self.enabled)
        out = {t: (float(vals[k]), int(cnt.get(t, 0))) for k, t in enumerate(self.enabled)}
        if reset:
            if acc is not None:
                acc.zero_()
            cnt.clear()
        return out

    def _objective(self, vectors, counts=None):
        order = [t for t in self.enabled if t in vectors]
        return ops.weighted_mean_sum([vectors[t] for t in order], [self.weights[t] for t in order],
                                     None if counts is None else [counts.get(t) for t in order])

    # ---- eager step -------------------------------------------------------------------------------------
    # Exact cross-rank graph-LayerNorm statistics (SURVEY 8e caveat 1, optional): with several ranks every graph LayerNorm
    # sums its segment statistics over the ranks (forward and backward, ops.set_graph_ln_exchange), so the step computes
    # what ONE process computes on the global batch -- the reference's semantics at that batch size -- instead of the
    # default per-rank statistics (each replica = the reference at its local batch size).  Six small collectives per step
    # on the compute stream.  On an RCCL group they are captured with the rest of the N-rank step (one graph incl. the
    # exchange, _capture_exchange_graph); on a group whose collectives cannot be captured (gloo) the mode steps eagerly.
    exact_graph_ln = False

    def _exact_ln_on(self) -> bool:
        return bool(self.exact_graph_ln and self.sync is not None and self.sync.world > 1)

    def _ln_exchange_scope(self):
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev = ops.set_graph_ln_exchange(self.sync.sum_small) if self._exact_ln_on() else ops.set_graph_ln_exchange(None)
            try:
                yield
            finally:
                ops.set_graph_ln_exchange(prev)
        return scope()

    def forward_backward(self, batches, merged=None):
        self.optimizer.zero_grad()
        if self.input_hook is not None:
            self.input_hook()
        self._scope_streams()
        prev = ops.set_wgrad_side_streams(self.wgrad_side_streams)
        prev_g = ops.set_wgrad_grouping(self.wgrad_grouping, self._wgrad_count(batches, merged))
        prev_d = ops.set_deferred_forks(self.deferred_forks)
        try:
            # (eagerly issued steps end with the same grouped tail launch as captured ones: same tile variants, same bits)
            self._install_tail(self._tail_only_plan([t for t in self.enabled if batches.get(t) is not None]))
            with self._ln_exchange_scope():
                total, vectors = self._backward_pass(batches, merged)
            ops.join_wgrad(force=True)
        finally:
            ops.set_last_wgrad_hook(None, None)
            ops.set_wgrad_side_streams(prev)
            ops.set_wgrad_grouping(prev_g)
            ops.set_deferred_forks(prev_d)
        return total, vectors

    def step(self, batches: Mapping[str, Data], merged: Optional[Data] = None):
        with STAGE_LOCK:
            if self._use_stages():
                out = self._staged_step(batches, merged)
            else:
                with self._learn_single_writers():
                    total, vectors = self.forward_backward(batches, merged)
                self._exchange_and_update()
                out = total.detach(), {t: v.detach() for t, v in vectors.items()}
            self._count_losses(out[1])
            return out

    # ---- gradient slots with one writer per step: stored, not cleared + accumulated (optim.FlatAdam.learn_begin / store_begin) ------
    # Learnt from the EAGER steps of every execution structure (one-piece and staged), applied to every CAPTURE (one rank, the
    # staged graphs of the N-rank step, the one-graph exchange): the gradient exchange reads the flat buffer region by region
    # once backward has finished it -- whether a slot was cleared and added into or stored makes no difference to what it reads.
    def _learn_single_writers(self):
        import contextlib
        opt = self.optimizer

        @contextlib.contextmanager
        def scope():
            learn = (getattr(opt, "materialised", False) and hasattr(opt, "learn_begin") and not torch.cuda.is_current_stream_capturing()
                     and switches.enabled("grad_store"))
            prev = ops.set_grad_slot_provider(opt.learn_begin()) if learn else None  # (which gradient slots have ONE writer per step)
            try:
                yield
            finally:
                if learn:
                    ops.set_grad_slot_provider(prev)
                    opt.learn_end()
        return scope()

    def _grad_store_begin(self):
        """Install the 'store' provider for a capture (None: off / nothing learnt); ``zero_flat_grads`` leaves the stored slots out
        until ``_grad_store_end``."""
        opt = self.optimizer
        if getattr(self, "_grad_store_off", False) or not hasattr(opt, "store_begin") or not switches.enabled("grad_store"):
            return None
        prov = opt.store_begin()
        if prov is None:
            return None
        return (ops.set_grad_slot_provider(prov),)

    def _grad_store_end(self, prev) -> None:
        if prev is not None:
            import sys
            ops.set_grad_slot_provider(prev[0])
            self.optimizer.store_end(ok=sys.exc_info()[0] is None)  # (checks that every slot left uncleared was written exactly once)
            self._grad_store_slots = len(self.optimizer.store_slots)

    def _join_zero(self):
        """Captured steps clear the gradient buffer on a side stream beside the forward pass: wait for it before the first
        launch that writes a gradient."""
        if getattr(self, "_zero_pending", False):
            ops.drain_deferred()
            torch.cuda.current_stream().wait_stream(self._zero_stream)
            self._zero_pending = False

    def _backward_pass(self, batches, merged=None):
        """Forward + backward of the objective (gradients accumulate into the parameters' slots): (objective, loss vectors)."""
        total, vectors, _ = self.losses(batches, merged)
        self._ride_loss_sums(vectors)
        self._join_zero()
        # (the seed of the objective's backward is a persistent 1.0: ``backward()`` without it fills a fresh tensor with a torch
        #  kernel on the chain, between the loss and the first backward launch)
        one = getattr(self, "_grad_one", None)
        if one is None or one.device != total.device or one.dtype != total.dtype:
            one = self._grad_one = torch.ones((), dtype=total.dtype, device=total.device)
        total.backward(gradient=one)
        return total, vectors

    # ---- staged backward: gradient exchange overlapped with the rest of backward --------------------------------------
    # With several ranks the flat gradient is exchanged REGION BY REGION as backward finishes it: the task heads'
    # region after stage A (forward + heads' backward), the SAGE stack's after stage B, the TRN's after stage C.  The
    # stages are pieces of ONE backward pass, cut at detached leaves (the backbone output per task, the TRN output),
    # so every gradient is what the single pass computes; each stage is its own hipGraph under capture and the
    # collectives run between the graph launches, on the communication stream, while the next stage computes.
    staged = None  # None: automatic (several ranks, flat buffers laid out [TRN | SAGE stack | heads]); True / False: forced

    def _stage_regions(self):
        return None  # steps that support stages return [(lo, hi) heads, (lo, hi) SAGE stack, (lo, hi) TRN]

    def _use_stages(self) -> bool:
        if self.staged is False or not getattr(self.optimizer, "materialised", False):
            return False
        if self.staged is None and (self.sync is None or self.sync.world <= 1):
            return False
        # (the sharded update is region-wise too: each region is reduce-scattered when backward has finished it, dist.GradSync.start)
        return self._stage_regions() is not None

    def _exchange_region(self, region):
        if self.sync is not None and self.sync.world > 1:
            # (handoff: the region's weight gradients may still be running on their side streams -- the communication stream
            #  waits for them, the backward stream goes straight on with the next stage)
            after = ops.take_wgrad_streams() if getattr(self, "_handoff", False) else ()
            self.sync.start(self.optimizer, *region, after=after)

    def _stage_join(self):
        """End of a backward stage: the stage's gradients are complete once the weight-gradient side streams are joined."""
        if not getattr(self, "_handoff", False):
            ops.join_wgrad(force=True)

    def _finish_staged(self):
        if self.sync is not None and self.sync.world > 1:
            self.sync.finish_and_step(self.optimizer)
        else:
            self.optimizer.step()

    def _staged_step(self, batches, merged=None):
        regions = self._stage_regions()
        if self.sync is not None:
            self.sync.begin_step()
        self.optimizer.zero_grad()
        if self.input_hook is not None:
            self.input_hook()
        self._scope_streams()
        prev = ops.set_wgrad_side_streams(self.wgrad_side_streams)
        prev_g = ops.set_wgrad_grouping(self.wgrad_grouping, self._wgrad_count(batches, merged))
        prev_d = ops.set_deferred_forks(self.deferred_forks)
        try:
            self._install_tail(self._tail_only_plan([t for t in self.enabled if batches.get(t) is not None]))
            with self._ln_exchange_scope(), self._learn_single_writers():
                total, vectors = self._stage_a(batches, merged)
                self._exchange_region(regions[0])
                self._stage_b()
                self._exchange_region(regions[1])
                self._stage_c()
                self._exchange_region(regions[2])
        finally:
            ops.set_last_wgrad_hook(None, None)
            ops.set_wgrad_side_streams(prev)
            ops.set_wgrad_grouping(prev_g)
            ops.set_deferred_forks(prev_d)
        self._finish_staged()
        return total.detach(), {t: v.detach() for t, v in vectors.items()}

    def _exchange_and_update(self):
        opt = self.optimizer
        if hasattr(opt, "materialised") and not opt.materialised:
            opt._materialise()
        if self.sync is not None and self.sync.world > 1:
            regions = self._stage_regions() if getattr(self.sync, "shard_update", False) else None
            if regions is not None:
                # sharded update: WHICH rank steps which slice must not depend on how the step was issued -- the moments of a slice
                # live on its rank -- so a one-piece backward (the first step, before the flat buffers exist) shards region by
                # region exactly as the staged steps that follow it do
                self.sync.begin_step()
                for region in regions:
                    self.sync.start(opt, *region)
                self.sync.finish_and_step(opt)
            else:
                self.sync.reduce_and_step(opt)  # chunked: Adam of chunk i overlaps the collectives of the later chunks
        else:
            opt.step()

    # ---- training loop entry: eager for the first steps, then the captured step on static-shape batches -------------
    use_graph = True
    graph_after = 2  # eager steps before the capture (they build the flat optimizer buffers and warm the allocator)

    def train_step(self, batches: Mapping[str, Data], merged: Optional[Data] = None):
        """One optimizer step on device batches, for training loops (main_temporal / main_egopack).

        The first ``graph_after`` steps run eagerly.  The next one is captured on private copies of its batches, and every
        later step whose batches have the SAME signature (``batch_signature``: shapes, dtypes, node counts, edge CAPACITIES
        -- the loaders deliver fixed-size batches of fixed-length sequences, so that is every step but a short last one,
        also for LTA batches whose edge count moves with the labels) copies its
        values into those buffers and replays the graph: the step the benchmark measures instead of ~150 launches issued
        from Python.  Anything else (different shapes, several ranks with an eager exchange path, ``use_graph`` off) takes
        the eager step.  Returns what ``step`` returns; after a replay the loss vectors are the graph's static outputs,
        valid until the next call.  ``loop_counts`` tallies replayed vs eager steps."""
        self._steps_seen = getattr(self, "_steps_seen", 0) + 1
        if not hasattr(self, "loop_counts"):
            self.loop_counts = {"replayed": 0, "eager": 0}  # per training loop; the entry points log and reset it per epoch
        st = getattr(self, "_train_static", None)
        ref = _shared_blob(batches, merged)
        if not (ref is not None and st is not None and self.use_graph and st.get("gsig") == ref.gsig and len(st.get("slots", ())) > 1
                and st["xkey"] == _feature_key(batches, merged)):
            ensure_features(batches, merged)  # (staged without their feature block, and this step is not a replay that gathers it)
        first = next(iter(batches.values()))
        on_device = first.x.is_cuda if torch.is_tensor(first.__dict__.get("x", None) if "x" in first.__dict__ else first.x) else (
            ref is not None and ref.dev.is_cuda)
        if (not self.use_graph or self._steps_seen <= self.graph_after or not on_device
                or (self._exact_ln_on() and not self._one_graph_exchange_ok())):
            self.loop_counts["eager"] += 1
            return self.step(batches, merged)
        if self.fused and len([t for t in self.enabled if batches.get(t) is not None]) > 1 and merged is None:
            self.loop_counts["eager"] += 1
            return self.step(batches, merged)  # (the caller did not stage a merged batch: nothing static to replay on)
        # batches out of ONE packed transfer (data.to_device_packed) with the layout the static buffers were built from: one
        # device-to-device copy of the byte buffer + one of the feature block instead of a copy per tensor, no signature walk
        if ref is not None and st is not None and st.get("gsig") == ref.gsig and st["xkey"] == _feature_key(batches, merged):
            slot = self._input_slot(st, ref, batches, merged)
            st["fast"] = None
            self.loop_counts["replayed"] += 1
            self._select_slot(slot)
            total = self.replay()
            slot["done"] = torch.cuda.current_stream().record_event()  # (this slot's buffers are read until here)
            return total.detach(), {t: v.detach() for t, v in self._static_out[1].items()}
        # equal structure fingerprints (stage_batches attaches them) + equal feature shapes = equal signature: the walk over
        # every tensor of the step's batches (0.2-0.3 ms of host time per step) is skipped for such a step
        fast = _fast_key(batches, merged)
        sig = st["sig"] if (st is not None and fast is not None and fast == st.get("fast")) else batch_signature(batches, merged)
        if st is None:
            blob = None
            if ref is not None:
                blob, static_b, static_m = _static_from_blob(ref, batches, merged)
            else:
                clone = lambda d: None if d is None else _clone_batch(d, pad_edges=True)
                static_b = {t: clone(b) for t, b in batches.items()}
                static_m = clone(merged)
                _rewire_packed(static_b, static_m)
                copy_batch_values(static_b, static_m, batches, merged)
            self.capture(static_b, static_m, warmup=0)
            st = self._train_static = {"sig": batch_signature(static_b, static_m), "batches": static_b, "merged": static_m,
                                       "fast": fast}
            if blob is not None:
                st.update(blob=blob, gsig=ref.gsig, xkey=_feature_key(batches, merged))
            st["slots"] = [self._slot_of_capture(blob, static_b, static_m)]
            if st["sig"] != sig:  # (cannot happen: the clones mirror the originals)
                self._train_static = None
                self.loop_counts["eager"] += 1
                return self.step(batches, merged)
        elif st["sig"] != sig:
            self.loop_counts["eager"] += 1
            return self.step(batches, merged)
        else:
            copy_batch_values(st["batches"], st["merged"], batches, merged, dst_walks=st.setdefault("walks", {}))
            if fast is not None and st.get("fast") != fast:
                st["fast"] = fast  # (the static buffers now hold this structure)
        self.loop_counts["replayed"] += 1
        self._select_slot(st["slots"][0])
        total = self.replay()
        st["slots"][0]["done"] = torch.cuda.current_stream().record_event()
        return total.detach(), {t: v.detach() for t, v in self._static_out[1].items()}

    # ---- two captured steps on alternating input buffers ------------------------------------------------------------------------------
    # A replay reads its batch from fixed buffers, and it reads the feature block at both ends of the step (first contraction, last
    # weight gradient): the next batch cannot be written there while the step runs, and written between two replays the copies (57 MB
    # of features + the index buffer) were 30 us on the stream both replays are ordered on.  A loop that feeds packed transfers
    # (data.to_device_packed) therefore captures the step TWICE, on two sets of input buffers, and alternates: while the replay on
    # one set runs, the next batch is copied into the other on a stream of its own (behind the event that ends the last replay that
    # read that set, and behind the staging of the batch).  The two graphs share everything else -- parameters, optimizer state,
    # gradient buffer, dropout offset word, loss accumulators -- and run one after the other on the training stream.
    double_buffered_inputs = None  # None: switches.enabled("double_buffered_inputs") for one-rank steps whose optimizer is in the graph
    gathers_inputs = False  # set once the step gathers the store rows into its own input buffers (see _input_slot)

    def _slot_of_capture(self, blob, static_b, static_m) -> dict:
        return {"graph": self._graph, "out": self._static_out, "static_in": getattr(self, "_static_in", None), "blob": blob,
                "batches": static_b, "merged": static_m, "done": None}

    def _select_slot(self, slot) -> None:
        self._graph, self._static_out, self._static_in = slot["graph"], slot["out"], slot["static_in"]

    def _double_buffer_ok(self) -> bool:
        on = self.double_buffered_inputs
        on = switches.enabled("double_buffered_inputs") if on is None else bool(on)
        return bool(on and getattr(self, "_fuse_adam", False) and not isinstance(self._graph, list))

    def _input_slot(self, st, ref, batches, merged) -> dict:
        """The set of input buffers this step's replay reads, holding this step's batch (see above)."""
        slots = st["slots"]
        cur = torch.cuda.current_stream()
        if self._double_buffer_ok():
            k = st["turn"] = (st.get("turn", 0) + 1) % 2
            if k == len(slots):  # the second capture, on private copies of THIS batch (they hold its values: nothing to fill)
                self._select_slot(slots[0])
                blob, static_b, static_m = _static_from_blob(ref, batches, merged)
                self.capture(static_b, static_m, warmup=0)
                slots.append(self._slot_of_capture(blob, static_b, static_m))
                return slots[k]
        else:
            k = 0
        slot = slots[k]
        side = cur
        if len(slots) > 1:
            if not hasattr(self, "_input_stream"):
                self._input_stream = torch.cuda.Stream(priority=-1 if switches.enabled("staging_priority") else 0)
            side = self._input_stream
            ready = getattr(ref, "ready", None)
            if ready is not None:
                side.wait_event(ready)  # (the staging of this batch: engine.StagedBatches)
            else:
                side.wait_stream(cur)  # (staged on the training stream by the caller)
            if slot["done"] is not None:
                side.wait_event(slot["done"])
        store = getattr(ref, "store", None)
        with torch.no_grad(), torch.cuda.stream(side):
            slot["blob"].copy_(ref.dev, non_blocking=True)
            if merged is not None:
                if torch.is_tensor(merged.__dict__.get("x")):
                    slot["merged"].x.copy_(merged.x, non_blocking=True)
                slot["merged"]._struct_key = getattr(merged, "_struct_key", 0)
            for t, b in batches.items():
                if b is not None:
                    dst = slot["batches"][t]
                    if not torch.is_tensor(b.__dict__.get("x")):
                        # staged without its feature block: the store rows (they arrived with the transfer) are gathered straight
                        # into this set's block -- one pass over the rows instead of a gather + a copy of the block
                        store.gather(b.__dict__["x_idx"], out=dst.x, dtype=dst.x.dtype)
                    elif merged is None:
                        dst.x.copy_(b.x, non_blocking=True)
                    dst._struct_key = getattr(b, "_struct_key", 0)
        if len(slots) > 1 and store is not None and switches.enabled("step_gathers_inputs"):
            self.gathers_inputs = True  # (from now on a StagedBatches that feeds this step leaves the gathers to it)
        if side is not cur:
            for v in (ref.dev, getattr(merged, "x", None), *(getattr(b, "x", None) for b in batches.values() if b is not None)):
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(side)
            cur.wait_event(side.record_event())
        return slot

    # ---- hipGraph capture ---------------------------------------------------------------------------------
    def capture(self, batches: Mapping[str, Data], merged: Optional[Data] = None, warmup: int = 2):
        """Capture forward+backward(+Adam if no gradient exchange) for THESE device tensors (static
        shapes and addresses: refill them in place between replays).

        The stored gradient slots (``_grad_store_begin``) are whatever the last EAGER step learnt; when the captured batches
        write them differently (another live-task set: MTL loaders of unequal length, per-task backbone passes) the provider or
        its end-of-capture check raises -- the capture is then taken once more with every slot cleared and accumulated, and
        ``capture_notes`` says so: a slower step, never a crash of the training loop and never a wrong gradient."""
        with STAGE_LOCK:  # (a staging thread makes no HIP call while a capture is open: StagedBatches._ahead)
            try:
                return self._capture_once(batches, merged, warmup)
            except RuntimeError as e:
                if not str(e).startswith("grad_store:") or getattr(self, "_grad_store_off", False):
                    raise
                if self.sync is not None and self.sync.world > 1 and self._one_graph_exchange_ok():
                    raise  # (a failed capture that holds collectives is not retried in this process: see one_graph_exchange)
                torch.cuda.synchronize()
                self._grad_store_off = True
                self.capture_notes = [*getattr(self, "capture_notes", []), f"stored gradient slots off for this step ({e})"]
                return self._capture_once(batches, merged, 0)

    def _capture_once(self, batches, merged, warmup):
        opt = self.optimizer
        self._loss_acc_for(next(self.model.parameters()).device)  # (the running loss sums' slots: allocated outside the capture)
        if self._exact_ln_on() and not self._one_graph_exchange_ok():
            raise RuntimeError("exact_graph_ln sums the graph-LayerNorm statistics over the ranks inside the step: this process "
                               "group's collectives cannot be captured in a hipGraph -- use step() / train_step() (eager) in this mode")
        live = [t for t in self.enabled if batches.get(t) is not None]
        if self.fused and len(live) > 1 and merged is None:  # index work must stay outside the capture
            merged = merge_batches([batches[t] for t in live]).to(batches[live[0]].pos.device)
        if warmup > 0 or not getattr(opt, "materialised", True):
            self._scope_streams()
            side = ops.unexcluded_stream()  # (a pooled handle that is not one of this step's head / task streams)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(warmup, 1)):  # also materialises the flat buffers
                    self.step(batches, merged)
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.sync is not None and self.sync.world > 1 and ((self._use_stages() and self._one_graph_exchange_ok()) or self._exact_ln_on()):
            # the eager steps above issued collectives and THIS capture records collectives (the opt-in one-graph exchange, the
            # exact cross-rank LayerNorm statistics): the process group's watchdog must have retired the eager ones before the
            # capture takes the communicator's stream into capture mode (dist.quiesce_before_capture: the round-3 abort).  The
            # staged graphs -- the N-rank default -- and the one-piece captures issue their collectives BETWEEN graph launches and
            # never capture that stream: they do not pay the settle time (a heuristic: one watchdog sweep, see the docstring).
            self.sync.quiesce()
        if hasattr(opt, "invalidate_lo_shadows") and not self._adam_keeps_lo():
            opt.invalidate_lo_shadows()  # (a captured step must contain every refresh of the low halves it relies on)
        # (when the Adam launch itself writes the low halves -- FlatAdam.adam_writes_lo, once they exist -- the warm-up steps left
        #  them fresh and every replay's Adam leaves them fresh for the next one: no split launch in the captured step)
        self._graph_has_exchange = False
        self._graph_adam_lo = False
        if self._use_stages():
            return self._capture_staged(batches, merged)
        fuse_adam = self.sync is None or self.sync.world <= 1
        g = torch.cuda.CUDAGraph()
        opt.sync_hyper_source()  # (the step constants are computed inside the graph from a device-side step counter)
        ops.rng_device_offset(opt.flat_p.device)  # (the dropout offset word exists BEFORE the capture: created inside, its fill is a node)
        self._hyper_in_graph = False
        self._scope_streams()
        prev = ops.set_wgrad_side_streams(self.wgrad_side_streams)
        prev_g = ops.set_wgrad_grouping(self.wgrad_grouping, self._wgrad_count(batches, merged))
        prev_d = ops.set_deferred_forks(self.deferred_forks)
        early = self._early_adam_plan(live) if fuse_adam else None
        # gradient slots with one writer per step (learnt from the eager steps above, FlatAdam.learn_begin) are stored, not cleared +
        # accumulated: the step's buffer clear shrinks to what is still added into (EGK_DISABLE=grad_store); also when the
        # optimizer follows a gradient exchange outside the graph
        store_prev = self._grad_store_begin()
        try:
            with torch.cuda.graph(g, stream=ops.unexcluded_stream(), capture_error_mode=CAPTURE_MODE):
                ops.stamp("step_start")
                # the gradient buffer is cleared BESIDE the forward pass (nothing writes a gradient before the first backward
                # launch): 100 MB of memset off the chain's head; joined in _join_zero() before backward starts
                hyper_here = fuse_adam and switches.enabled("hyper_in_graph")
                self._hyper_in_graph = hyper_here
                if not switches.enabled("zero_stream"):
                    opt.zero_flat_grads()
                    if hyper_here:
                        opt.prepare_hyper(in_capture=True)
                else:
                    if not hasattr(self, "_zero_stream"):
                        self._zero_stream = torch.cuda.Stream()

                    def issue_zero(ev):  # (behind the forward pass's first launch: see ops.defer_after_next_launch)
                        self._zero_stream.wait_event(ev)
                        with torch.cuda.stream(self._zero_stream):
                            opt.zero_flat_grads()
                            if hyper_here:  # the step's Adam constants: one thread, beside the forward pass
                                opt.prepare_hyper(in_capture=True)
                    if not switches.enabled("zero_deferred"):
                        issue_zero(torch.cuda.current_stream().record_event())
                    else:
                        ops.defer_after_next_launch(issue_zero)
                    self._zero_pending = True
                self._rng_in_graph = False
                if self.input_hook is not None:
                    self.input_hook()
                if early is None and not fuse_adam:
                    self._install_tail(self._tail_only_plan(live))
                if early is not None:
                    early["rng"] = switches.enabled("rng_in_graph") and switches.enabled("rng_early")
                    ops.set_last_wgrad_hook(early["param"], early["hook"], pre=self._join_gradient_branches)
                    ops.set_graphone_backward_hook(early.get("graphone_hook"))
                    if early["tail"]:
                        g0 = opt.flat_g.data_ptr()
                        ops.set_last_wgrad_tail(g0 + 4 * early["lo"], g0 + 4 * early["hi"])
                # (no join with the weight-gradient side stream when a backward() call returns: the gradients are read below, behind
                #  ``join_wgrad(force=True)``)
                total, vectors = self._backward_pass(batches, merged)
                self._join_zero()  # (a backward path that did not: the memset must at least precede the optimizer)
                ops.set_last_wgrad_hook(None, None)
                ops.join_wgrad(force=True)
                ops.stamp("backward_done")
                # the Philox offset word of the dropout launches moves on INSIDE the graph (beside nothing that reads it: every
                # dropout launch of the step is done when the optimizer starts): replay k draws the masks of offset base + k * stride
                # without a separate launch in front of every replay -- and without a launch of its own: it rides in an optimizer
                # launch (egk_adam_step_bump)
                rng_here = switches.enabled("rng_in_graph") and not (early is not None and early.get("rng_done"))
                bump = (ops.rng_device_offset(opt.flat_p.device), ops.RNG_DEVICE_STRIDE) if rng_here else None
                if fuse_adam:
                    if early is not None and early["fired"]:
                        torch.cuda.current_stream().wait_stream(early["stream"])
                        opt.launch(None, early["lo"], early["hi"], bump=bump if early["hi"] > early["lo"] else None)
                        bump = None if early["hi"] > early["lo"] else bump
                    elif early is not None and early.get("done"):  # (a slice was stepped, the last weight gradient's hook never ran)
                        torch.cuda.current_stream().wait_stream(early["stream"])
                        for a, b in _minus([(0, opt.flat_p.numel())], early["done"]):
                            opt.launch(None, a, b, bump=bump)
                            bump = None
                    else:
                        opt.launch(bump=bump)
                        bump = None
                    ops.stamp("adam_done")
                if switches.enabled("rng_in_graph"):
                    if bump is not None:
                        ops.advance_rng_device(opt.flat_p.device)
                    self._rng_in_graph = True
        finally:
            ops.set_last_wgrad_hook(None, None)
            ops.set_graphone_backward_hook(None)
            ops.set_wgrad_side_streams(prev)
            ops.set_wgrad_grouping(prev_g)
            ops.set_deferred_forks(prev_d)
            self._grad_store_end(store_prev)
        self._graph, self._static_out, self._fuse_adam = g, (total, vectors), fuse_adam
        self._graph_adam_lo = bool(fuse_adam and self._adam_keeps_lo())  # (the captured Adam launches write the low halves)
        # the graph holds raw addresses: keep every tensor it reads alive for as long as the graph exists
        # (in particular the merged batch -- CSR arrays, positions, segment pointers -- when it was built here)
        self._static_in = (batches, merged)
        return g

    # captured single-GPU step: start Adam on everything but the last weight gradient's slots beside that launch.  Only
    # for steps that have JOINED every other gradient producer into the backward stream by then (MTLStep with its
    # head-wise backward does; a step whose branches are still running on their own streams at that point must not)
    early_adam = False

    def _tail_only_plan(self, live):
        """(first TRN weight, flat range of the temporal pooling's slots) when the step can end with ONE grouped launch of the
        pooling's weight gradients (ops.set_last_wgrad_tail) although no optimizer slice runs beside it -- the steps whose
        Adam follows a gradient exchange.  None when the layout does not allow it."""
        opt = self.optimizer
        tp = getattr(self.model, "temporal_pooling", None)
        first = getattr(tp, "proj", [None])[0] if tp is not None else None
        if (first is None or not hasattr(opt, "region_of") or not getattr(opt, "materialised", False) or not (self.fused or len(live) == 1)
                or not switches.enabled("tail_group") or not self.headwise_backward_ok()
                or first.weight.numel() >= 512 * 128 * 128):  # (a wide first linear's weight gradient stays a launch of its own)
            return None
        params = [p for p in tp.parameters() if p.requires_grad]
        lo, hi = opt.region_of(params)
        slots = [opt._slot_of[id(p)] for p in params if id(p) in opt._slot_of]
        if not (hi > lo and lo % 8 == 0 and hi % 8 == 0 and sum(n for _, n in slots) == hi - lo):
            return None
        return first.weight, lo, hi

    def headwise_backward_ok(self) -> bool:
        return bool(getattr(self, "headwise_backward", False))

    def _install_tail(self, plan, hook=None) -> None:
        if plan is None:
            return
        g0 = self.optimizer.flat_g.data_ptr()
        ops.set_last_wgrad_hook(plan[0], hook or (lambda: None))
        ops.set_last_wgrad_tail(g0 + 4 * plan[1], g0 + 4 * plan[2])

    def _early_adam_plan(self, live):
        """The step ends with two launches that have the chip to themselves one after the other: the weight gradient of
        the first TRN linear (the last node of backward, 58 GFLOP, compute-bound) and Adam (750 MB, HBM-bound).  When the
        backbone runs once per step (fused pass or a single task) every other gradient is final when that weight gradient
        is launched: Adam over the rest of the flat buffers starts beside it on its own stream, Adam over the slots of
        that weight and bias follows it.  Elementwise: the same update, bit for bit."""
        opt = self.optimizer
        tp = getattr(self.model, "temporal_pooling", None)
        first = getattr(tp, "proj", [None])[0] if tp is not None else None
        if not (self._early_adam_ok() and first is not None and hasattr(opt, "region_of") and (self.fused or len(live) == 1)):
            return None
        total = opt.flat_p.numel()

        def region(params):
            lo, hi = opt.region_of(params)
            slots = [opt._slot_of[id(p)] for p in params if id(p) in opt._slot_of]
            ok = hi > lo and lo % 8 == 0 and hi % 8 == 0 and sum(n for _, n in slots) == hi - lo  # (adjacent slots)
            return (lo, hi) if ok else None
        # the late Adam slice: the whole temporal pooling when its slots are one block (then the step's tail is ONE grouped
        # launch of its weight gradients, ops.set_last_wgrad_tail), else the first linear alone
        # (a first linear whose weight gradient alone is >= 512 tiles of 128 x 128 -- the shipped pooling width 4096: 1152 -- fills
        #  the chip by itself: its launch stays alone and the late slice is that layer's, so that Adam over the rest of the pooling
        #  runs beside it; Hp = 4096 step 2.785-2.796 -> 2.770-2.772 ms.  At Hp = 1024 (288 tiles) the pooling's three weight
        #  gradients are one grouped launch, 123 against 120 + 66 us.)
        tail = None
        if switches.enabled("tail_group") and first.weight.numel() < 512 * 128 * 128:
            tail = region([p for p in tp.parameters() if p.requires_grad])
        reg = tail or region([p for p in (getattr(first, "weight", None), getattr(first, "bias", None)) if p is not None])
        if reg is None:
            return None
        lo, hi = reg
        if not hasattr(self, "_adam_stream"):
            self._adam_stream = torch.cuda.Stream()
        plan = {"param": first.weight, "lo": lo, "hi": hi, "stream": self._adam_stream, "fired": False, "tail": tail is not None}
        def hook():
            if plan["fired"]:
                return
            main = torch.cuda.current_stream()

            def issue(ev):
                # (issued AFTER the last weight gradient's own launch, behind an event recorded here: see
                #  ops.defer_after_next_launch -- the launch created first keeps the backward stream's hardware queue)
                side = ops.wgrad_side_stream(main)
                plan["stream"].wait_event(ev)
                if side is not None:
                    plan["stream"].wait_stream(side)
                with torch.cuda.stream(plan["stream"]):
                    # the dropout offset word moves on here, beside the last weight gradient (every dropout launch of the step
                    # is long done), INSIDE the first optimizer launch: no launch of its own
                    bump = (ops.rng_device_offset(opt.flat_p.device), ops.RNG_DEVICE_STRIDE) if plan.get("rng") else None
                    for a, b in _minus([(0, lo), (hi, total)], plan.get("done", ())):  # (``done``: slices stepped earlier in the step)
                        opt.launch(None, a, b, bump=bump)
                        bump = None
                    if bump is not None:  # (no slice left to step here: the word still moves on)
                        ops.advance_rng_device(opt.flat_p.device)
            if plan.get("rng"):
                plan["rng_done"] = True  # (every dropout launch of the step has been issued: this is backward's end)
            ops.defer_after_next_launch(issue)
            plan["fired"] = True
        plan["hook"] = hook
        return plan

    # (measured and not kept: Adam over the SAGE + head slots started when backward enters the temporal pooling, beside the
    #  TRN backward chain: 1.85 vs 1.75 ms -- the HBM-bound launch slows the chain more than the shorter tail saves)

    def _early_adam_ok(self) -> bool:
        return bool(self.early_adam)

    def _gradient_branch_streams(self):
        """Streams on which parts of backward may still be running when the LAST weight gradient of the step is launched."""
        return []

    def _join_gradient_branches(self) -> None:
        """Called on the backward stream right before the step's last weight gradient (ops.set_last_wgrad_hook ``pre``): the
        branches of a one-call backward that run on other streams are joined HERE -- what they parked for the grouped
        weight-gradient launches (operands made on their streams) is issued next, and the optimizer slice that starts beside the
        last weight gradient reads every gradient they wrote.  (Autograd itself joins them only when backward() returns.)"""
        main = torch.cuda.current_stream()
        for st in self._gradient_branch_streams():
            with torch.cuda.stream(st):
                live_branch = torch.cuda.is_current_stream_capturing()
            if live_branch and st != main:
                main.wait_stream(st)

    # The N-rank step as ONE hipGraph: the three stages, the region-wise collectives between them (communication stream, forked
    # and joined inside the capture) and the per-chunk Adam launches.  Three graph launches + ~13 collectives + ~13 Adam
    # launches issued from Python per step become one graph launch (--exchange-dry-run 8 on one GPU: 1.69 -> 1.53 ms).
    # RCCL collectives enqueue device work only and are capturable; the process group's watchdog keeps polling its events from
    # its own thread, legal under CAPTURE_MODE 'thread_local'.
    #
    # OPT-IN.  The default for several ranks is the path with the most evidence: three staged graphs with the collectives
    # issued eagerly between the graph launches (two real rank processes run it, tools/two_rank_check.py).  This mode has only
    # ever met a 1-rank RCCL group, so it is taken when the caller asks for it -- the attribute, or EGK_ENABLE=one_graph_exchange
    # -- typically after a pre-flight child process per rank has captured and replayed it with exit code 0 (bench.py
    # ``one_graph_probe``).  A capture that holds collectives and FAILS is not retried in this process: the exception reaches
    # the caller (after a failed capture nothing guarantees that the stream / communicator is reusable; the round-3 catch-and-
    # continue onto the staged graphs is gone), who ends the attempt -- the parent picks the next mode for ALL ranks.
    one_graph_exchange = False

    def _one_graph_exchange_ok(self) -> bool:
        forced = switches.override("one_graph_exchange")  # (EGK_ENABLE / EGK_DISABLE win over the attribute)
        want = self.one_graph_exchange if forced is None else forced
        return bool(want and self.sync is not None and self.sync.world > 1 and self.sync.capturable())

    def _capture_exchange_graph(self, batches, merged):
        opt, sync = self.optimizer, self.sync
        regions = self._stage_regions()
        live = [t for t in self.enabled if batches.get(t) is not None]
        g = torch.cuda.CUDAGraph()
        opt.grad_scale = 1.0 / sync.world
        opt.sync_hyper_source()
        self._hyper_in_graph = True
        count = opt.step_count
        self._scope_streams()
        prev = ops.set_wgrad_side_streams(self.wgrad_side_streams)
        prev_g = ops.set_wgrad_grouping(self.wgrad_grouping, self._wgrad_count(batches, merged))
        prev_d = ops.set_deferred_forks(self.deferred_forks)
        sync.begin_step()
        sync.hyper_ready = True
        self._handoff = switches.enabled("wgrad_handoff")
        prev_h = ops.set_wgrad_handoff(self._handoff)
        store_prev = self._grad_store_begin()  # (single-writer gradient slots are stored: the collectives read final values either way)
        try:
            with torch.cuda.graph(g, stream=ops.unexcluded_stream(), capture_error_mode=CAPTURE_MODE):
                # (the gradient buffer is cleared beside the forward pass, as in the one-rank capture: joined by _join_zero()
                #  before the heads' backward writes the first gradient)
                if not hasattr(self, "_zero_stream"):
                    self._zero_stream = torch.cuda.Stream()

                def issue_zero(ev):
                    self._zero_stream.wait_event(ev)
                    with torch.cuda.stream(self._zero_stream):
                        opt.zero_flat_grads()
                        opt.prepare_hyper(in_capture=True)  # (the step's Adam constants, from the device-side step counter)
                ops.stamp("step_start")
                ops.defer_after_next_launch(issue_zero)
                self._zero_pending = True
                self._rng_in_graph = False
                if self.input_hook is not None:
                    self.input_hook()
                ln_scope = self._ln_exchange_scope()  # (exact cross-rank graph-LN statistics: their collectives are captured too)
                ln_scope.__enter__()
                total, vectors = self._stage_a(batches, merged)
                self._join_zero()
                ops.stamp("heads_done")
                self._exchange_region(regions[0])
                # beside the step's LAST weight-gradient launch (the grouped tail of the temporal pooling): the Adam slices of
                # the regions exchanged by then, each behind its collective -- what the one-rank capture does with its early
                # Adam launch; only the pooling's own slices are left for the end of the step
                fired = []

                def early_adam_hook():
                    if fired or not switches.enabled("exchange_early_adam"):
                        return
                    fired.append(True)
                    if not hasattr(self, "_adam_stream"):
                        self._adam_stream = torch.cuda.Stream()

                    def issue(ev):
                        self._adam_stream.wait_event(ev)
                        sync.step_started_chunks(opt, self._adam_stream)
                    ops.defer_after_next_launch(issue)
                self._install_tail(self._tail_only_plan(live), early_adam_hook if self._early_adam_ok() else None)
                self._stage_b()
                ops.stamp("stack_done")
                self._exchange_region(regions[1])
                self._stage_c()
                ln_scope.__exit__(None, None, None)
                ops.stamp("backward_done")
                self._exchange_region(regions[2])
                sync.finish_and_step(opt)
                ops.stamp("adam_done")
                if switches.enabled("rng_in_graph"):
                    ops.advance_rng_device(opt.flat_p.device)
                    self._rng_in_graph = True
        finally:
            self._handoff = False
            ops.set_wgrad_handoff(prev_h)
            sync.hyper_ready = False
            sync.begin_step()
            opt.step_count = count  # (finish_and_step counted the capture; replay() counts the steps that run)
            ops.set_last_wgrad_hook(None, None)
            ops.set_wgrad_side_streams(prev)
            ops.set_wgrad_grouping(prev_g)
            ops.set_deferred_forks(prev_d)
            self._grad_store_end(store_prev)
        self._graph, self._static_out, self._fuse_adam = g, (total, vectors), True
        self._graph_adam_lo = self._adam_keeps_lo()  # (every chunk's captured Adam launch writes the low halves of its slice)
        self._graph_has_exchange = True
        self._static_in = (batches, merged, self._stage_state, self._cuts)
        return g

    def _capture_staged(self, batches, merged):
        """Three graphs (one per backward stage) from one memory pool; the gradient exchange sits between them."""
        self._graph_has_exchange = False
        if self._one_graph_exchange_ok():
            return self._capture_exchange_graph(batches, merged)  # (an exception ends the attempt: see one_graph_exchange)
        opt = self.optimizer
        gs = [torch.cuda.CUDAGraph() for _ in range(3)]
        self._rng_in_graph = False  # (no staged graph advances the Philox offset word: replay() does, also after a one-piece capture)
        self._scope_streams()
        prev = ops.set_wgrad_side_streams(self.wgrad_side_streams)
        prev_g = ops.set_wgrad_grouping(self.wgrad_grouping, self._wgrad_count(batches, merged))
        prev_d = ops.set_deferred_forks(self.deferred_forks)
        store_prev = self._grad_store_begin()  # (one provider over the three captures: every learnt slot is written in exactly one of them)
        try:
            live = [t for t in self.enabled if batches.get(t) is not None]
            cap = ops.unexcluded_stream()  # (one capture stream for the three graphs, never a registered head / task stream)
            with torch.cuda.graph(gs[0], stream=cap, capture_error_mode=CAPTURE_MODE):
                # the gradient buffer is cleared BESIDE the forward pass, as in the one-rank capture (a fork and a join inside
                # the first graph; _stage_a joins it before the heads' backward writes the first gradient)
                if not switches.enabled("zero_stream"):
                    opt.zero_flat_grads()
                else:
                    if not hasattr(self, "_zero_stream"):
                        self._zero_stream = torch.cuda.Stream()

                    def issue_zero(ev):
                        self._zero_stream.wait_event(ev)
                        with torch.cuda.stream(self._zero_stream):
                            opt.zero_flat_grads()
                    ops.defer_after_next_launch(issue_zero)
                    self._zero_pending = True
                if self.input_hook is not None:
                    self.input_hook()
                total, vectors = self._stage_a(batches, merged)
                self._join_zero()  # (a stage A that did not: the fork must end inside this graph)
            pool = gs[0].pool()
            with torch.cuda.graph(gs[1], pool=pool, stream=cap, capture_error_mode=CAPTURE_MODE):
                self._install_tail(self._tail_only_plan(live))  # (stage B ends with the stack's flush, stage C is the tail launch)
                self._stage_b()
            with torch.cuda.graph(gs[2], pool=pool, stream=cap, capture_error_mode=CAPTURE_MODE):
                self._stage_c()
        finally:
            ops.set_last_wgrad_hook(None, None)
            ops.set_wgrad_side_streams(prev)
            ops.set_wgrad_grouping(prev_g)
            ops.set_deferred_forks(prev_d)
            self._grad_store_end(store_prev)
        self._graph, self._static_out, self._fuse_adam = gs, (total, vectors), False
        self._static_in = (batches, merged, self._stage_state, self._cuts)  # everything the graphs read stays alive
        return gs

    def replay(self):
        """One training step from the captured graph(s)."""
        self._count_losses(self._static_out[1])
        opt = self.optimizer
        if not getattr(self, "_rng_in_graph", False):
            ops.advance_rng_device(opt.flat_p.device)
        if isinstance(self._graph, list):
            regions = self._stage_regions()
            if self.sync is not None:
                self.sync.begin_step()
            for g, region in zip(self._graph, regions):
                g.replay()
                self._exchange_region(region)
            self._finish_staged()
            return self._static_out[0]
        if self._fuse_adam:
            if getattr(self, "_hyper_in_graph", False):
                opt.sync_hyper_source()  # (host -> device only when lr / grad_scale changed or the step count was set)
                self._graph.replay()
                opt.note_captured_step()
            else:
                opt.prepare_hyper()
                self._graph.replay()
            opt.step_count += 1
            # the graph's Adam launch moved flat_p on the DEVICE without opt.launch() running on the host: every low half a
            # 'bf16x3' contraction marked fresh before this replay (validation's precise pass between two epochs) is stale now --
            # unless the captured Adam launches write the low halves themselves: then ALL of them are fresh behind a replay
            if getattr(self, "_graph_adam_lo", False):
                opt._lo_fresh = [(0, opt.flat_p.numel())]
            else:
                opt.invalidate_lo_shadows()
        else:
            self._graph.replay()
            self._exchange_and_update()  # (one-piece graph, the exchange and the optimizer behind it)
        return self._static_out[0]


class MTLStep(StepBase):
    """One multi-task pre-training step (BASELINE configs 2, 3, 5; reference main_temporal.train :49-134)."""

    def __init__(self, model, tasks: Mapping[str, torch.nn.Module], criteria: Mapping[str, torch.nn.Module],
                 weights: Mapping[str, float], optimizer, fused_backbone: bool = True, sync: Optional[GradSync] = None,
                 parallel_heads: bool = True):
        self._init_base(model, tasks, weights, optimizer, fused_backbone, sync, parallel_heads)
        self.criteria = dict(criteria)

    def _head(self, t: str, feat, d):
        task = self.tasks[t]
        f = task.forward_features(feat)
        logits = task.forward_logits(f, d) if t == "oscc" else task.forward_logits(f)
        return self.criteria[t](logits, d.y), logits

    def losses(self, batches: Mapping[str, Data], merged: Optional[Data] = None):
        feats = self.features(batches, merged)
        vectors, logits_out = self._run_heads(feats, lambda t, feat: self._head(t, feat, batches[t]))
        return self._objective(vectors), vectors, logits_out

    # the projection heads of the task batches as ONE chain of grouped launches (ops.grouped_projection) instead of one
    # chain per task on its own stream: same shapes, different rows, different weights -- 3 x 2048-row launches that each
    # fill half of the chip become one 6144-row launch per stage (bf16 mode, once the optimizer's flat buffers exist)
    grouped_heads = True

    # a head whose classifier has ONE logit under BCE-with-logits (PNR) runs classifier + loss + their gradients as one row
    # pass (ops.linear1_bce) instead of ten launches of [N, H] x [H, 1] matrix work: same-box A/B of the headline step
    # 1.650 -> 1.624 ms.  Same values up to summation order (a row reduction instead of an MFMA K walk), so the bitwise
    # equalities between execution structures (tests/test_gpu_dist.py) are stated with it off
    one_pass_heads = True

    def _one_pass_head_ok(self, t: str) -> bool:
        from .criterion import BCEWithLogitsNone
        return (self.one_pass_heads and hasattr(self.tasks[t], "fused_head_loss") and type(self.criteria[t]) is BCEWithLogitsNone
                and switches.enabled("rowdot_head"))

    def _one_pass_oscc_ok(self, t: str) -> bool:
        """The OSCC head (max pool -> 2-logit classifier -> cross entropy, one loss element per SEQUENCE) as pool + one launch
        (ops.linear2_ce): its eleven short launches sat on the critical path of the 4-task step between forward and backward."""
        from .criterion import CrossEntropyNone
        return (t == "oscc" and self.one_pass_heads and hasattr(self.tasks[t], "fused_head_loss")
                and type(self.criteria[t]) is CrossEntropyNone and getattr(self, "_fused_loss", True)
                and switches.enabled("rowdot_head") and switches.enabled("oscc_one_pass"))

    grouped_classifiers = True

    def _banked_tasks(self, order, proj_leaves):
        """Tasks whose multi-head classifier banks can share grouped launches (ops.grouped_classifier_banks): at least two."""
        if not self.grouped_classifiers or not switches.enabled("grouped_classifiers"):
            return []
        cand = []
        for t in order:
            task = self.tasks[t]
            cls = getattr(task, "classifiers", None)
            if cls is None or (task.training and any(c[0].p > 0 for c in cls)):
                continue  # (no multi-head bank, or classifier dropout active: the per-task path)
            if getattr(cls[0][1].weight, "_egk_bank_views", None) is None or not isinstance(self.criteria[t], MetricSelectorWrapper):
                continue
            cand.append(t)
        if len(cand) < 2:
            return []
        views = [self.tasks[t].classifiers[0][1].weight._egk_bank_views for t in cand]
        return cand if ops.grouped_classifier_banks_ok([proj_leaves[t] for t in cand], views) else []

    def _heads_forward_backward(self, feats):
        """Heads' forward AND backward, every head inside its own stream context, on detached copies of the backbone
        features: returns (objective, loss vectors, {task: leaf}) with d(objective)/d(features) in ``leaf.grad``.
        The objective is sum_t w_t * mean(v_t) (main_temporal.py:99-128), so head t's backward starts from the
        constant w_t / numel(v_t) -- the value the one-call backward of the objective hands it, bit for bit."""
        leaves = {t: f.detach().requires_grad_(True) for t, f in feats.items()}
        batches = self._head_batches
        if not hasattr(self, "_coef_grads"):
            self._coef_grads = {}
        order = list(leaves)
        nets = [self.tasks[t].net for t in order]
        # Row compaction (exact): the heads are row-wise and a node whose labels are all ``ignore_index`` has loss 0 and gradient 0
        # (AR labels the centre node of a sequence only, data/ego4d_fho.py:222-223): such a task's projection, classifiers, loss and
        # their backward run on its labelled rows (data.live_label_rows, padded with all-zero rows / ignored labels to a multiple of
        # 64) and the feature gradient goes back to full height with zero rows elsewhere.  Loss vectors keep one element per node.
        live = {t: batches[t] for t in order if self._compact_head_ok(t, batches[t], leaves[t])}
        n_full = {t: leaves[t].shape[0] for t in order}
        labels = {t: (live[t].live_y if t in live else batches[t].y) for t in order}
        compact_v = {}  # the compacted heads' loss vectors as computed (the objective sums these; n_full divides)

        def full_vector(t, v):
            if t not in live:
                return v.detach()
            # one element per node for the caller (meters); the launch feeds nothing on the chain: it rides with the parked weight
            # gradients' next flush, on their stream
            compact_v[t] = v.detach()
            out = torch.empty(n_full[t], dtype=v.dtype, device=v.device)
            inv, src = live[t].live_inv, compact_v[t]
            ops.park_rider(lambda: ops.expand_rows(src, inv, out=out), (src, inv, out))
            return out
        # (a task whose labelled rows are an arithmetic progression -- one labelled node per sequence -- hands the grouped
        #  projection its full-height features and a row spec: a strided view stands for the gathered rows, no launch)
        specs, grouped = {}, False
        if self.grouped_heads and len(order) > 1:
            specs = {t: (*live[t].live_ap, live[t].live_inv) for t in live if getattr(live[t], "live_ap", None) is not None}
            grouped = ops.grouped_projection_ok([leaves[t] for t in order], nets, [specs.get(t) for t in order])
            if not grouped:
                specs = {}
        heads_in = {t: (leaves[t] if (t not in live or t in specs) else ops.live_rows(leaves[t], live[t].live_idx, live[t].live_inv))
                    for t in order}
        if self.grouped_heads and len(order) > 1 and not grouped:
            grouped = ops.grouped_projection_ok([heads_in[t] for t in order], nets)
        proj = proj_leaves = None
        if grouped:
            proj = ops.grouped_projection([heads_in[t] for t in order], nets, specs=[specs.get(t) for t in order])
            proj_leaves = {t: f.detach().requires_grad_(True) for t, f in zip(order, proj)}
            ops.stamp("heads_proj_fwd_done")
        else:
            specs = {}

        def head(t, leaf):
            # AR / LTA: one loss element per node, back-propagated below with the constant w_t / numel -- known before the
            # loss is computed, so the cross entropy emits its gradient in the same launch (ops.loss_seed)
            # (the loss vector has one element per NODE, so the seed divides by the full node count also for a compacted head)
            n_loss = n_full[t] if (t in ("ar", "lta", "pnr") and getattr(self, "_fused_loss", True)) else 0
            oscc_one = self._one_pass_oscc_ok(t) and batches[t].y.dim() == 1
            if oscc_one:
                n_loss = int(batches[t].y.numel())  # one loss element per sequence
            with ops.loss_seed(self.weights[t] / n_loss if n_loss else None):
                task, d = self.tasks[t], batches[t]
                f = leaf if grouped else task.forward_features(leaf)  # grouped: ``leaf`` is the projected feature block
                one_pass = None
                if oscc_one:
                    one_pass = task.fused_head_loss(f, d, d.y, getattr(self.criteria[t], "label_smoothing", 0.0))
                elif n_loss and self._one_pass_head_ok(t):
                    one_pass = task.fused_head_loss(f, d.y)  # one-logit classifier + BCE + their gradients: one row pass
                if one_pass is not None:
                    v, logits = one_pass
                elif oscc_one:  # (did not apply after all: the contraction path WITHOUT an announced seed, as before)
                    with ops.loss_seed(None):
                        logits = task.forward_logits(f, d)
                        v = self.criteria[t](logits, d.y)
                    n_loss = 0
                else:
                    logits = task.forward_logits(f, d) if t == "oscc" else task.forward_logits(f)
                    v = self.criteria[t](logits, labels[t])
            n_out = v.numel() if t not in live else n_full[t]  # elements of the loss vector the objective averages over
            if n_loss and n_out != n_loss:
                raise RuntimeError(f"head {t}: {n_out} loss elements where {n_loss} were announced to the fused loss")
            if v.numel():
                # the constant the objective's backward hands this head (w_t / numel): one tensor per task, filled once
                key = (t, v.numel(), n_out, v.dtype, v.device)
                g = self._coef_grads.get(key)
                if g is None or g._egk_coef != self.weights[t] / n_out:
                    g = self._coef_grads[key] = torch.full_like(v, self.weights[t] / n_out, dtype=v.dtype).detach()
                    g._egk_coef = self.weights[t] / n_out
                v.backward(gradient=g)
            return full_vector(t, v), logits
        # the multi-head classifier banks of several tasks (AR and LTA: same widths, own rows and weights) as ONE chain of
        # grouped launches on the main stream -- grouped classifier contraction, one fused cross entropy per task, grouped dX --
        # instead of one chain per task on its own stream (each fork / join of a stream costs more than these launches)
        banked = self._banked_tasks(order, proj_leaves) if grouped else []
        banked_vectors = {}

        def banked_chain():
            views = [self.tasks[t].classifiers[0][1].weight._egk_bank_views for t in banked]
            n_loss = {t: n_full[t] for t in banked}  # (one loss element per NODE: a compacted head's seed divides by the full count)
            all_logits = ops.grouped_classifier_banks([proj_leaves[t] for t in banked], views,
                                                      fused_loss=getattr(self, "_fused_loss", True))
            vs, gs = [], []
            multi = None
            if getattr(self, "_fused_loss", True) and switches.enabled("ce_multi"):
                # the cross entropies of the banked tasks as ONE launch (each writes its own loss vector and gradient operand)
                sel = [self.criteria[t].select(logits, labels[t]) for t, logits in zip(banked, all_logits)]
                if len({s_[2] for s_ in sel}) == 1:
                    multi = ops.cross_entropy_multi([(s_[0], s_[1]) for s_ in sel], [self.weights[t] / n_loss[t] for t in banked],
                                                    sel[0][2])
            for i, (t, logits) in enumerate(zip(banked, all_logits)):
                if multi is not None:
                    v = multi[i]
                else:
                    with ops.loss_seed(self.weights[t] / n_loss[t] if getattr(self, "_fused_loss", True) else None):
                        v = self.criteria[t](logits, labels[t])
                if v.numel() != proj_leaves[t].shape[0]:
                    raise RuntimeError(f"head {t}: {v.numel()} loss elements for {proj_leaves[t].shape[0]} rows")
                key = (t, v.numel(), n_loss[t], v.dtype, v.device)
                g = self._coef_grads.get(key)
                if g is None or g._egk_coef != self.weights[t] / n_loss[t]:
                    g = self._coef_grads[key] = torch.full_like(v, self.weights[t] / n_loss[t], dtype=v.dtype).detach()
                    g._egk_coef = self.weights[t] / n_loss[t]
                vs.append(v)
                gs.append(g)
                banked_vectors[t] = full_vector(t, v)
            torch.autograd.backward(vs, gs)  # ONE pass: both cross entropies, then the grouped banks' backward once
        with ops.bank_grad_handoff():  # every head's logits feed exactly one loss node here
            src = proj_leaves if grouped else heads_in
            rest = {t: f for t, f in src.items() if t not in banked}
            vectors, _ = self._run_heads(rest, head, main_job=banked_chain if banked else None)
            vectors.update(banked_vectors)
            vectors = {t: vectors[t] for t in order if t in vectors}
        if grouped:  # back on the main stream: the grouped projection's backward down to the backbone features
            ops.stamp("heads_classifiers_done")
            live = [t for t in order if proj_leaves[t].grad is not None]
            if live:
                outs = dict(zip(order, proj))
                torch.autograd.backward([outs[t] for t in live], [proj_leaves[t].grad for t in live])
        with torch.no_grad():
            # the reported objective feeds nothing on the chain (every head's backward started from its known seed): its one-workgroup
            # reduction rides with the parked weight gradients' next flush, on their stream, instead of sitting between the heads'
            # backward and the backbone's
            src = {t: compact_v.get(t, v) for t, v in vectors.items()}
            cnt = {t: (n_full[t] if t in compact_v else None) for t in vectors}
            if not switches.enabled("objective_rider") or not src:
                total = self._objective(src, cnt)
                self._ride_loss_sums(src)
            else:
                # (every enabled task has its slot in the launch -- absent ones add exactly 0 -- so that the running per-task loss
                #  sums, ``loss_sums``, accumulate in the same launch at fixed positions)
                order = list(self.enabled)
                total = torch.empty((), dtype=torch.float32, device=next(iter(src.values())).device)
                acc = self._loss_acc_for(total.device)
                ops.park_rider(lambda: ops.weighted_mean_sum_into(total, [src.get(t) for t in order], [self.weights[t] for t in order],
                                                                  [cnt.get(t) for t in order], acc=acc),
                               (total, *[src[t] for t in order if t in src]))
        return total, vectors, leaves

    compact_heads = True  # heads on the labelled rows only (data.live_label_rows); EGK_DISABLE=compact_heads: every row

    def _compact_head_ok(self, t: str, d, leaf) -> bool:
        """Task ``t``'s head runs on its labelled rows: a multi-head cross-entropy head (MetricSelectorWrapper over
        CrossEntropyNone with ignore_index -1: the row's loss and gradient are exactly zero when every head's label is -1),
        no active dropout in the head (its masks are drawn by row position), the batch built by data.collate with its
        ``live_*`` index arrays on the features' device."""
        if not self.compact_heads or not switches.enabled("compact_heads") or t not in ("ar", "lta"):
            return False
        idx = getattr(d, "live_idx", None)
        if idx is None or getattr(d, "live_inv", None) is None or getattr(d, "live_y", None) is None:
            return False
        if not (idx.is_cuda and idx.device == leaf.device and d.live_inv.numel() == leaf.shape[0]):
            return False
        from .criterion import CrossEntropyNone
        crit = self.criteria.get(t)
        if not isinstance(crit, MetricSelectorWrapper) or type(getattr(crit, "criterion", None)) is not CrossEntropyNone:
            return False
        # (a row labelled only in a column the wrapper does not select is kept: its loss and gradient are still exactly zero)
        task = self.tasks[t]
        drops = [m for m in task.modules() if isinstance(m, torch.nn.Dropout) or type(m).__name__ == "Dropout"]
        return not (task.training and any(getattr(m, "p", 0) > 0 for m in drops))

    headwise_backward = True  # False: one backward() call over all streams (kept for A/B measurements)
    early_adam = True

    def _early_adam_ok(self) -> bool:  # the heads are joined into the main stream before the backbone's backward starts
        return bool(self.early_adam and self.headwise_backward)

    def _backward_pass(self, batches, merged=None):
        if not self.headwise_backward:
            return super()._backward_pass(batches, merged)
        self._head_batches = batches
        feats = self.features(batches, merged)
        ops.stamp("fwd_backbone_done")
        self._join_zero()
        total, vectors, leaves = self._heads_forward_backward(feats)
        ops.stamp("heads_done")
        if switches.enabled("heads_flush"):
            # the heads' parked weight gradients (classifier banks, projections) go out as one launch NOW, beside the first
            # links of the backbone's dX chain; left parked, the backbone's first weight gradient would flush nine problems
            # as 8 + 1
            ops.flush_wgrad(in_backward=False, force=True)
        order = [t for t in feats if leaves[t].grad is not None]
        torch.autograd.backward([feats[t] for t in order], [leaves[t].grad for t in order])
        return total, vectors

    # ---- the three backward stages (StepBase._staged_step) ---------------------------------------------------------
    def _stage_regions(self):
        opt, model = self.optimizer, self.model
        if not hasattr(opt, "region_of") or not hasattr(model, "net") or getattr(model, "temporal_pooling", None) is None:
            return None
        head_params = [p for t in self.enabled for p in self.tasks[t].parameters()]
        mid_params = list(model.net.parameters())
        mid_ids = {id(p) for p in mid_params}
        trn_params = [p for p in model.parameters() if id(p) not in mid_ids]
        heads, mid, trn = opt.region_of(head_params), opt.region_of(mid_params), opt.region_of(trn_params)
        total = opt.flat_p.numel()
        tiles = sorted(r for r in (trn, mid, heads) if r[1] > r[0])
        ok = tiles and tiles[0][0] == 0 and tiles[-1][1] == total and all(a[1] == b[0] for a, b in zip(tiles, tiles[1:]))
        return [heads, mid, trn] if ok else None  # (any other layout: the one-piece backward + pipelined exchange)

    def _stage_a(self, batches, merged):
        """Forward, heads' forward / backward: final gradients of the head parameters + d(objective)/d(features)."""
        self._cuts = []

        def cut(x):
            leaf = x.detach().requires_grad_(True)
            self._cuts.append((x, leaf))
            return leaf
        self.model.stage_cut = cut
        try:
            feats = self.features(batches, merged)
        finally:
            self.model.stage_cut = None
        self._head_batches = batches
        self._join_zero()  # (a capture that clears the gradient buffer beside the forward pass: before the first gradient)
        total, vectors, leaves = self._heads_forward_backward(feats)
        self._stage_join()
        self._stage_state = (feats, leaves)
        return total, vectors

    def _stage_b(self):
        """Backbone output -> TRN output: final gradients of the SAGE stack."""
        feats, leaves = self._stage_state
        order = list(feats)
        torch.autograd.backward([feats[t] for t in order], [leaves[t].grad for t in order])
        self._stage_join()

    def _stage_c(self):
        """TRN output -> inputs: final gradients of the temporal pooling."""
        if self._cuts:
            torch.autograd.backward([x for x, _ in self._cuts], [leaf.grad for _, leaf in self._cuts])
            self._stage_join()


def _minus(ranges, holes):
    """The element ranges ``ranges`` without the ranges ``holes`` (lists of [lo, hi))."""
    out = list(ranges)
    for h0, h1 in holes:
        nxt = []
        for a, b in out:
            if h1 <= a or b <= h0:
                nxt.append((a, b))
                continue
            if a < h0:
                nxt.append((a, h0))
            if h1 < b:
                nxt.append((h1, b))
        out = nxt
    return [(a, b) for a, b in out if b > a]


class EgoPackStep(StepBase):
    """One novel-task step of main_egopack.train with late fusion (BASELINE configs 4, 5; reference
    main_egopack.py:45-159): backbone (train/eval mode and grad mode as configured) -> primary projection; aux
    projections DETACHED -> GraphONE.interact -> fused logits -> primary.compute_loss."""

    order = ("ar", "oscc", "lta", "pnr")  # order of the loss terms in main_egopack.train
    # grouped weight gradients and late forks: OFF until round 4 (3.77-3.89 against 3.62-3.72 ms then); with the OSCC head as six
    # launches and the three GraphONE backward chains on three queues they measure 3.20-3.24 against 3.47-3.48 ms
    wgrad_grouping_default = True
    deferred_forks_default = True
    wgrad_group_count = 8  # 2048-row batches: eight H x H problems = 512 tiles = two per CU (3.16 -> 3.09 ms; six on the 3-task step)
    AUX_ORDER = {"ar": ("lta", "oscc", "pnr"), "oscc": ("ar", "lta", "pnr"), "lta": ("ar", "oscc", "pnr"),
                 "pnr": ("ar", "oscc", "lta")}  # main_egopack.py:121-147

    def __init__(self, model, tasks: Mapping[str, torch.nn.Module], graphone, weights: Mapping[str, float], optimizer,
                 backprop_temporal_graph: bool = True, temporal_graph_train_mode: bool = False,
                 sync: Optional[GradSync] = None, fused_backbone: bool = True, parallel_heads: bool = True):
        self._init_base(model, tasks, weights, optimizer, fused_backbone, sync, parallel_heads)
        self.graphone = graphone
        self.backprop, self.train_mode = backprop_temporal_graph, temporal_graph_train_mode
        # Tasks that are not trained here only lend their projection heads to the detached auxiliary features (reference
        # main_egopack.py:53: ``.detach()``): their parameters never receive a gradient and the reference's Adam skips them
        # (grad is None).  They are marked frozen, which changes no result and lets the operand copies of their weights (bf16
        # copies, bf16 halves of the three-product path) be made once instead of at every use.
        for t, task in self.tasks.items():
            if t not in self.enabled:
                for p in task.parameters():
                    p.requires_grad_(False)

    # The nearest-prototype search is an INDEX op (reference graphONE.py:119-141: argsort of the cosine distances): its lists must
    # be the reference's, whatever the storage type of the training pass.  With bf16 activations the detached auxiliary features
    # carry the rounding of a dozen bf16 layers (~1 % relative), enough to flip ~5 % of the neighbour positions at K = 4096.  In
    # the bf16 modes they therefore come from a second, forward-only pass over the same batch in 'bf16x3' mode (f32 activations,
    # every contraction as three bf16 products of split operands: f32-grade at ~3x the bf16 cost, ops.precise_scope): backbone
    # + auxiliary projections, no gradient -- the reference detaches these features (main_egopack.py:53).  The GraphONE stages
    # consume the same values (rounded to the activation type).
    precise_search = True
    precise_stream = True  # the precise pass on its own stream beside the training pass's forward (False: in line, A/B)
    # Adam over everything but the temporal pooling's slots beside the step's last weight-gradient launch (StepBase._early_adam_plan):
    # the step is ONE backward() call whose last node is the first TRN linear -- every other node has been issued by then, on
    # the backward stream or on the branch streams below, which the optimizer slice waits for.  Config 4: 55 M parameters,
    # Adam alone at the end of the step lasted 261 us of 2.84 ms
    early_adam = True

    def _early_adam_ok(self) -> bool:
        return bool(self.early_adam and self.backprop and switches.enabled("early_adam"))

    def _gradient_branch_streams(self):
        return [*getattr(self.graphone, "_task_streams", ()), *getattr(self, "_head_streams", ())]

    def _early_adam_plan(self, live):
        """+ GraphONE's own slice (EGK_DISABLE=graphone_adam: off): its stage parameters (half of the step's parameters in config
        4) have their final gradients when the grouped GraphONE backward returns (ops.set_graphone_backward_hook) -- Adam over
        that slice then runs beside the backbone's backward instead of in the step's tail.  Round 4: the tail shrank from 275 to
        114 us and the backbone's backward grew by as much (2.746-2.751 against 2.72-2.80 ms, tools/round4/c4_g1adam_ab.sh: the
        memory-bound launch slows the chain it runs beside) -- opt-in then.  Round 5, with the searches grouped and the precise pass
        first: 2.42-2.45 against 2.475-2.478 ms (three alternating rounds): the default.  Elementwise, the same update bit for bit
        (tests/test_gpu_configs.py::test_config4_graphone_optimizer_slice_is_the_same_update)."""
        plan = super()._early_adam_plan(live)
        opt = self.optimizer
        if plan is None or not switches.enabled("graphone_adam") or len(live) != 1:
            return plan
        params = [p for p in self.graphone.parameters() if p.requires_grad and id(p) in opt._slot_of]
        if not params:
            return plan
        g0, g1 = opt.region_of(params)
        slots = [opt._slot_of[id(p)] for p in params]
        if not (g1 > g0 and g0 % 8 == 0 and g1 % 8 == 0 and sum(n for _, n in slots) == g1 - g0 and (g1 <= plan["lo"] or g0 >= plan["hi"])):
            return plan
        plan["done"] = []

        def graphone_done():
            if plan["fired"] or plan["done"]:
                return
            ops.flush_wgrad(force=True)  # (what the interaction left parked; issued behind the backward stream's next launch)
            main = torch.cuda.current_stream()

            def issue(ev):
                side = ops.wgrad_side_stream(main)
                plan["stream"].wait_event(ev)
                if side is not None:
                    plan["stream"].wait_stream(side)
                with torch.cuda.stream(plan["stream"]):
                    opt.launch(None, g0, g1)
            ops.defer_after_next_launch(issue)
            plan["done"].append((g0, g1))
        plan["graphone_hook"] = graphone_done
        return plan

    def _precise_on(self) -> bool:
        return bool(self.precise_search and ops.get_compute() in ("bf16", "bf16_f32act") and switches.enabled("precise_search"))

    def _aux_names(self, primary: str):
        return [t for t in self.AUX_ORDER[primary] if t in self.graphone.task_labels]

    @torch.no_grad()
    def precise_aux_features(self, batches, merged=None, rng_snap=None, tape=None):
        """{primary: {aux task: f32 [N, H]}}: the auxiliary projections of every enabled task batch from the 'bf16x3' pass.
        ``tape`` (a list): the backbone's nodes leave their results in it (ops.dual_record) for the one-pass step."""
        opt = self.optimizer
        live = [t for t in self.enabled if batches.get(t) is not None]
        with ops.precise_scope(), ops.rng_replay(ops.rng_snapshot() if rng_snap is None else rng_snap):
            if getattr(opt, "materialised", False):
                opt.refresh_lo_shadows(list(self.model.parameters()))  # the backbone's low halves: one launch
            if tape is not None:
                with ops.dual_record() as rec:
                    feats = self.features(batches, merged)
                tape.extend(rec)
            else:
                feats = self.features(batches, merged)
            out = {}
            for t in live:
                others = self._aux_names(t)
                # the auxiliary projections of one batch as three grouped launches (+ one operand split) instead of ~5 per task at
                # the END of the precise chain, which the GraphONE stages wait for
                grouped = ops.grouped_projection_infer(feats[t], [self.tasks[o].net for o in others], out_f32=True) if len(others) > 1 else None
                out[t] = (dict(zip(others, grouped)) if grouped is not None
                          else {o: self.tasks[o].forward_features(feats[t], out_f32=True) for o in others})
        return out

    one_pass = True  # the bf16 training graph from the precise pass's results instead of a second backbone pass (see _one_pass_ok)

    def _one_pass_ok(self, batches, merged) -> bool:
        """ONE backbone pass per step (ops.dual_record / dual_replay) applies: a single task batch with bf16 features, gradients
        through the backbone, no active dropout in it (the keep masks of the two passes would have to be shared), statistics
        local to the rank.  EGK_DISABLE=one_pass: the two-pass step of rounds 3-5."""
        if not (self.one_pass and self.backprop) or not switches.enabled("one_pass") or not switches.enabled("one_pass"):
            return False
        live = [batches[t] for t in self.enabled if batches.get(t) is not None]
        if len(live) != 1 or isinstance(live[0].x, (list, tuple)) or live[0].x.dtype != torch.bfloat16 or ops.graph_ln_exchange_on():
            return False
        if ops.get_compute() != "bf16" or getattr(self.model, "stage_cut", None) is not None:
            return False
        drops = [m for m in self.model.modules() if type(m).__name__ == "Dropout" or isinstance(m, torch.nn.Dropout)]
        if self.model.training and any(getattr(m, "p", 0) > 0 for m in drops):
            return False
        tp = getattr(self.model, "temporal_pooling", None)
        return not (self.model.training and float(getattr(tp, "dropout", 0) or 0) > 0)

    def _search_ahead(self, precise) -> None:
        """The prototype searches of the coming ``GraphONE.interact`` calls on the CURRENT stream (the precise pass's): they need
        the precise features only, so they start when that pass ends -- not behind the join with the training pass's forward chain,
        which (created second) ends later: profiles/r05_c4_replay_timeline.txt vs r05b: the searches 907 -> 7xx us."""
        self.graphone.drop_searched()
        if not switches.enabled("search_ahead"):
            return
        for d in precise.values():
            self.graphone.search_ahead(d)

    def task_loss(self, primary: str, feat, data, aux_in=None, f_primary=None):
        task = self.tasks[primary]
        others = self._aux_names(primary)
        if f_primary is None:
            f_primary = task.forward_features(feat)
        if aux_in is None:
            with torch.no_grad():
                # f32 out of the projections' last contraction: the nearest-prototype search (an index op) ranks the f32
                # accumulators in every compute mode; GraphONE brings them to the activation type for its stages
                grouped = None
                if switches.enabled("grouped_aux"):
                    grouped = ops.grouped_projection_infer(ops.to_act(feat), [self.tasks[t].net for t in others], out_f32=True)
                aux_in = (dict(zip(others, grouped)) if grouped is not None
                          else {t: self.tasks[t].forward_features(feat, out_f32=True) for t in others})
        aux, closest = self.graphone.interact(aux_in)
        ops.stamp("stages_done")
        if (primary == "oscc" and getattr(task, "loss_func", None) == "ce" and hasattr(task, "fused_head_loss") and data.y.dim() == 1
                and switches.enabled("oscc_one_pass") and switches.enabled("rowdot_head")):
            # the four 2-logit classifiers (primary + one per auxiliary task) behind their max pools, the logit fusion, the loss and
            # every gradient as ONE launch: the objective is sum_t w_t mean(loss_t), so the loss vector's backward seed is the
            # constant w / B (ops.loss_seed) -- ~35 short launches of the contraction path otherwise
            with ops.loss_seed(self.weights[primary] / max(int(data.y.numel()), 1)):
                one = task.fused_head_loss(f_primary, data, data.y, smoothing=0.1, aux_features=aux,
                                           aux_streams=getattr(self.graphone, "stream_of", None))
            if one is not None:
                return one[0], one[1], aux, closest
        if primary == "oscc":
            logits = task.forward_logits(features=f_primary, batch=data, aux_features=aux)
        else:
            logits = task.forward_logits(features=f_primary, batch=getattr(data, "batch", None), aux_features=aux)
        return task.compute_loss(logits, data.y), logits, aux, closest

    def losses(self, batches: Mapping[str, Data], merged: Optional[Data] = None):
        self.model.train(self.train_mode)
        for t in self.tasks.values():
            t.train(True)
        self.graphone.train()
        snap = ops.rng_snapshot()
        precise, side = {}, None
        tape, gate_ev = None, {}
        first = next((b for b in batches.values() if b is not None), None)
        on_side = (self._precise_on() and self.precise_stream and first is not None and first.pos.is_cuda
                   and switches.enabled("precise_stream"))
        # Which chain is CREATED first keeps the launch queue under capture: the precise pass IS the step's critical chain and goes
        # first (2.529-2.549 against 2.565-2.586 ms with the training pass's forward chain created first, four alternating rounds).
        if self._precise_on():
            if on_side:
                # the precise pass is a chain of ~50 launches over the same few thousand rows as the training pass's forward:
                # forked onto its own stream, the two chains run side by side (each alone leaves most of the chip idle)
                main = torch.cuda.current_stream()
                if getattr(self, "_precise_side", None) is None:
                    self._precise_side = torch.cuda.Stream()  # (a high-priority stream: 5.05-5.26 against 2.53-2.55 ms in the captured step)
                    ops.exclude_wgrad_streams([self._precise_side])
                side = self._precise_side
                side.wait_stream(main)
                gate = os.environ.get("EGK_TRAIN_AFTER", "")  # (development: the training pass starts behind this phase of the precise pass)
                gate_ev = {}
                if gate:
                    ops.phase_callbacks({gate: lambda: gate_ev.setdefault("ev", side.record_event())})
                tape = [] if self._one_pass_ok(batches, merged) else None
                with torch.cuda.stream(side):
                    try:
                        precise = self.precise_aux_features(batches, merged, rng_snap=snap, tape=tape)
                    finally:
                        if gate:
                            ops.phase_callbacks(None)
                    ops.stamp("precise_done")
                    self._search_ahead(precise)
                    ops.stamp("search_done")
            else:
                precise = self.precise_aux_features(batches, merged, rng_snap=snap)
        import contextlib
        # (the two passes draw the same dropout offsets: same keep masks when the backbone is in train mode)
        if side is not None and switches.debug("serial_precise"):  # (measurement: the two passes one after the other)
            torch.cuda.current_stream().wait_stream(side)
        if side is not None and gate_ev.get("ev") is not None:
            torch.cuda.current_stream().wait_event(gate_ev["ev"])
        tape = tape if side is not None else None
        if tape:
            # ONE backbone pass: the training graph is built from the precise pass's taped results (ops.dual_replay) -- its nodes
            # launch roundings, no contractions -- behind the precise backbone, beside its auxiliary projections and the searches
            cur = torch.cuda.current_stream()
            for _, ts in tape:  # (every node of the replay waits for its own taped node: ops._tape_take)
                for t in ts.values():
                    if torch.is_tensor(t):
                        t.record_stream(cur)
            with ops.dual_replay(tape), torch.set_grad_enabled(self.backprop):
                feats = self.features(batches, merged)
        else:
            with (ops.rng_replay(snap) if self._precise_on() else contextlib.nullcontext()), torch.set_grad_enabled(self.backprop):
                feats = self.features(batches, merged)
        # the primary projection heads need nothing of the precise pass: issued BEFORE the join with its stream, so that they run
        # beside its tail instead of behind it (profiles/r04_c4_replay_timeline.txt: 896-990 us, 95 us in which nothing else ran)
        f_prim = {}
        if side is not None and len(feats) == 1 and switches.enabled("primary_early"):
            f_prim = {t: self.tasks[t].forward_features(f) for t, f in feats.items()}
        ops.stamp("train_fwd_done")
        if side is not None:
            main.wait_stream(side)
            ops.stamp("precise_joined")
            for d in precise.values():
                for a in d.values():
                    a.record_stream(main)
            for st in self.graphone.ahead_streams():
                main.wait_stream(st)  # (the searches forked off ``side``; joined HERE, into the origin stream: GraphONE.search_ahead)
            for a in self.graphone.searched_tensors():
                a.record_stream(main)

        def head(t, feat):
            loss, logits, _, _ = self.task_loss(t, feat, batches[t], aux_in=precise.get(t), f_primary=f_prim.get(t))
            return loss, logits
        vectors, logits_out = self._run_heads(feats, head)
        return self._objective(vectors), vectors, logits_out
