"""Data parallelism: one process per GPU, gradients exchanged with RCCL (torch.distributed backend
"nccl" on ROCm) over xGMI.

The reference is single-process; this is new.  The batch is sharded by rank
(data.BatchLoader), banks and parameters are replicated, and the only data-path collective is the
gradient all-reduce.  Gradients live in ONE flat fp32 buffer (optim.FlatAdam), so the exchange is a
few large all-reduces of contiguous slices launched on a side stream; the 1/world_size average is
folded into the Adam kernel's grad_scale.  Chunks are issued in REVERSE buffer order: the flat
buffer is laid out backbone-first, heads-last, and backward produces head gradients first."""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import switches


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group
    when WORLD_SIZE > 1.  backend defaults to 'nccl' (= RCCL) on GPU, 'gloo' on CPU."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def free_port() -> int:
    """A TCP port nobody listens on right now (bound to port 0 and released)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def init_single_rank_group(backend: str = "nccl") -> None:
    """A 1-rank process group for the exchange DRY RUN (``exchange_dry_run`` / ``--exchange-dry-run``: the N-rank code path
    on one GPU): its own rendezvous on a free port -- a fixed one collides with whatever else runs on the box."""
    if dist.is_initialized():
        return
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    for attempt in range(4):  # (a port found free can be taken before the store binds it: another port then)
        os.environ["MASTER_PORT"] = str(free_port())
        try:
            dist.init_process_group(backend, rank=0, world_size=1)
            return
        except Exception as e:  # noqa: BLE001  (DistNetworkError is not importable on every build)
            if "address already in use" not in str(e).lower() or attempt == 3:
                raise


def quiesce_before_capture(group=None, settle_s: float = 1.0) -> None:
    """Let the RCCL process group's WATCHDOG retire every collective issued so far before a hipGraph capture starts.

    Root cause of the round-3 driver abort (reproduced 3 times in 9 on fresh boxes, tools/round4/repro_abort.sh; the C++ trace is
    in profiles/r04_abort_root_cause.txt): ProcessGroupNCCL's watchdog thread keeps a copy of every EAGERLY issued collective
    and polls its end event (hipEventQuery) every ~100 ms until it has completed.  ``capture(warmup=...)`` issues eager steps
    -- with their collectives -- and starts capturing right behind them; when a capture that contains collectives then takes
    the communicator's stream into capture mode while the watchdog still holds one of those eager works, its next poll fails
    with hipErrorCapturedEvent ("operation not permitted on an event last recorded in a capturing stream"), the watchdog
    rethrows on its own thread and std::terminate aborts the process -- hundreds of milliseconds later, wherever the main
    thread happens to be.  A device synchronisation makes every pending work complete; one watchdog sweep later they are gone.
    No-op without an RCCL group.  The settle time is a HEURISTIC -- one watchdog sweep (~100 ms period) is assumed to finish within
    ``settle_s``; the process group offers no call that waits for its watchdog's list to drain -- so only captures that record
    collectives pay it (engine.StepBase.capture), and those are opt-in behind a per-rank probe process (bench.one_graph_probe)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    try:
        if dist.get_backend(group) != "nccl":
            return
    except Exception:  # noqa: BLE001  (not a member of the group)
        return
    import time
    torch.cuda.synchronize()
    time.sleep(settle_s)


def _via_host(t: torch.Tensor, group) -> bool:
    """A device tensor on a gloo group: ranks that share one GPU (the two-process test topology of
    tools/two_rank_check.py; RCCL refuses two ranks on one device) exchange through host memory."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum_(t: torch.Tensor, group=None) -> None:
    """Sum over the ranks in place, ordered on the current stream (RCCL), or synchronously through the host (gloo)."""
    if _skip["collectives"]:
        return
    if _via_host(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


_skip = {"collectives": False}  # measurement only (GradSync.skip_collectives): the exchange path without its collectives


def chunk_bounds(n: int, chunk_elems: int) -> List[tuple]:
    """[(begin, end)] covering [0, n) in chunks of ``chunk_elems`` (last one shorter), reverse order."""
    out = [(b, min(n, b + chunk_elems)) for b in range(0, n, chunk_elems)]
    return out[::-1]


class GradSync:
    """All-reduce(sum) of a flat gradient buffer in large chunks.

    On GPU the collectives are enqueued on a side stream that waits for the event recorded after
    backward, so the first chunks overlap whatever the compute stream does next, and the optimiser
    waits for ``done`` before it reads the buffer."""

    def __init__(self, world_size: int, chunk_mb: float = 32.0, group=None, compress: str = "none",
                 shard_update: Optional[bool] = None):
        """compress='bf16' (GPU only): the flat f32 gradient is converted to a bf16 copy by one kernel launch, the
        bf16 copy is all-reduced (half the bytes on every xGMI link) and the Adam kernel reads it directly.
        shard_update (default: EGK_ENABLE=sharded_update): reduce-scatter -> Adam on this rank's 1 / world slice ->
        all-gather of the parameters (``_sharded_step``) instead of all-reduce -> the whole Adam pass on every rank."""
        if compress not in ("none", "bf16"):
            raise ValueError(compress)
        self.world, self.group, self.compress = world_size, group, compress
        self.shard_update = (switches.enabled("sharded_update")) if shard_update is None else bool(shard_update)
        self.chunk_elems = max(8, (int(chunk_mb * (1 << 20) / 4) + 7) // 8 * 8)  # chunk starts stay 32-byte aligned
        self._side = None
        self._g16 = None
        self._inflight = []
        # region-wise exchange (start / finish_and_step): every chunk's Adam slice may be issued behind its collective on the
        # communication stream, beside the backward stages that follow (a region's parameters are not read again by them).
        # OPT-IN (EGK_ENABLE=adam_behind_collective or the attribute): the Adam writes (f32 parameters, bf16 shadows) then run
        # concurrently with the remaining backward graphs, and that ordering has never executed against real RCCL peers --
        # the default issues every slice after the last stage, behind its collective's event
        self.adam_behind_collective = switches.enabled("adam_behind_collective")
        self.hyper_ready = False  # set by a caller that has prepared the step's Adam constants itself (a captured exchange)

    def quiesce(self) -> None:
        """Before a hipGraph capture: see ``quiesce_before_capture``."""
        quiesce_before_capture(self.group)

    def capturable(self) -> bool:
        """Whether the exchange can be recorded into a hipGraph: RCCL collectives enqueue device work only (a gloo group moves
        the data through host memory, synchronously)."""
        return dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    @property
    def skip_collectives(self) -> bool:
        """Measurement only (bench.py): with this on, the gradient all-reduces are left out of the exchange path -- every rank
        steps on its LOCAL gradient, everything else of the path (conversion, stream hand-offs, per-chunk Adam) runs.  The
        step time with it minus the step time without is what the collectives cost beyond what backward hides."""
        return _skip["collectives"]

    @skip_collectives.setter
    def skip_collectives(self, on: bool) -> None:
        _skip["collectives"] = bool(on)

    def broadcast_(self, flat: torch.Tensor, src: int = 0):
        if self.world > 1:
            if _via_host(flat, self.group):
                h = flat.cpu()
                dist.broadcast(h, src=src, group=self.group)
                flat.copy_(h)
            else:
                dist.broadcast(flat, src=src, group=self.group)

    def sum_small(self, buf: torch.Tensor) -> None:
        """In-place sum over the ranks of a small tensor ON THE CURRENT STREAM (the statistics of a graph LayerNorm in the
        exact cross-rank mode, ops.set_graph_ln_exchange): the launches that follow on this stream read the result."""
        if self.world > 1:
            all_reduce_sum_(buf, self.group)

    def all_reduce_(self, flat_g: torch.Tensor) -> torch.Tensor:
        """Sum over ranks; returns the buffer that holds the summed gradient (``flat_g`` itself, or its bf16 copy
        under compression) once the work is ENQUEUED (GPU) or done (CPU/gloo)."""
        if self.world <= 1:
            return flat_g
        if self.compress == "bf16" and flat_g.is_cuda:
            from . import _lib
            from .ops import _ck, _p, _stream
            if self._g16 is None or self._g16.numel() != flat_g.numel():
                self._g16 = torch.empty(flat_g.numel(), dtype=torch.bfloat16, device=flat_g.device)
            _ck(_lib.load().egk_cast(_stream(), _p(flat_g), 0, _p(self._g16), 1, flat_g.numel()), "egk_cast")
            self._reduce_chunks(self._g16)
            return self._g16
        self._reduce_chunks(flat_g)
        return flat_g

    def _reduce_chunks(self, flat_g: torch.Tensor):
        bounds = chunk_bounds(flat_g.numel(), self.chunk_elems)
        if flat_g.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat_g.device)
            main = torch.cuda.current_stream(flat_g.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                for b, e in bounds:
                    all_reduce_sum_(flat_g[b:e], self.group)
            main.wait_stream(self._side)
        else:
            for b, e in bounds:
                all_reduce_sum_(flat_g[b:e], self.group)


    # ---- region-wise exchange for the staged backward (engine.StepBase): start() as soon as a region of the flat
    # gradient is final, finish_and_step() after the last stage ---------------------------------------------------------
    def begin_step(self) -> None:
        """Start of a staged step: forget whatever an ABANDONED step (an exception between two stages) left in flight, so that
        the first ``start`` of this step prepares the Adam constants and the coverage check counts this step's chunks only."""
        self._inflight.clear()

    def start(self, opt, lo: int, hi: int, after=()) -> None:
        """Enqueue conversion (compute stream) + all-reduce (communication stream) of flat_g[lo:hi] in chunks; the
        compute stream goes on with the next backward stage while the collectives run.  ``after``: streams whose work the
        region's gradients also come from (weight-gradient side streams the compute stream has NOT waited for)."""
        if hi <= lo:
            main = torch.cuda.current_stream(opt.flat_g.device)
            for s_ in after:
                main.wait_stream(s_)
            return
        flat_g = opt.flat_g
        if self.shard_update:
            return self._start_sharded(opt, lo, hi, after)
        compress = self.compress == "bf16"
        src = flat_g
        if compress:
            from . import _lib
            from .ops import _ck, _p, _stream
            if self._g16 is None or self._g16.numel() != flat_g.numel():
                self._g16 = torch.empty(flat_g.numel(), dtype=torch.bfloat16, device=flat_g.device)
            src = self._g16
        if self._side is None:
            self._side = torch.cuda.Stream(device=flat_g.device)
        main = torch.cuda.current_stream(flat_g.device)
        if not self._inflight and not self.hyper_ready:  # first region of this step: the step's Adam constants, ordered before every slice below
            opt.grad_scale = 1.0 / self.world
            opt.prepare_hyper()
        if compress:
            for s_ in after:  # (the conversion reads the gradients on the compute stream)
                main.wait_stream(s_)
        else:
            for s_ in after:
                self._side.wait_stream(s_)
        for b in range(lo, hi, self.chunk_elems):
            e = min(hi, b + self.chunk_elems)
            if compress:
                _ck(_lib.load().egk_cast(_stream(), _p(flat_g[b:e]), 0, _p(src[b:e]), 1, e - b), "egk_cast")
            ready = torch.cuda.Event()
            ready.record(main)
            self._side.wait_event(ready)
            with torch.cuda.stream(self._side):
                all_reduce_sum_(src[b:e], self.group)
                # the Adam slice of the chunk right behind its collective, on the communication stream: the parameters of a
                # region are not read again by the backward stages that follow it (heads: used in stage A only; SAGE stack:
                # not by the temporal pooling's backward), so the update runs beside those stages instead of after them
                if self.adam_behind_collective:
                    opt.launch(src, b, e)
                ev = torch.cuda.Event()
                ev.record(self._side)
            self._inflight.append({"b": b, "e": e, "ev": ev, "src": src, "stepped": self.adam_behind_collective})

    def step_started_chunks(self, opt, stream) -> None:
        """The Adam slice of every chunk started so far and not stepped yet, on ``stream``, each behind its collective: the
        engine calls this beside the LAST weight-gradient launch of backward (the parameters of the regions exchanged by then
        -- heads, SAGE stack -- are not read again), so that only the last region's slices are left for the end of the step."""
        for c in self._inflight:
            if not c["stepped"]:
                stream.wait_event(c["ev"])
                with torch.cuda.stream(stream):
                    if "shard" in c:
                        self._step_shard(opt, c)
                    else:
                        opt.launch(c["src"], c["b"], c["e"])
                    ev = torch.cuda.Event()
                    ev.record(stream)
                c["ev"], c["stepped"] = ev, True

    def finish_and_step(self, opt) -> None:
        """Adam launch per exchanged chunk, in the order the chunks were started, each behind its collective."""
        main = torch.cuda.current_stream(opt.flat_g.device) if opt.flat_g.is_cuda else None
        covered = 0
        regions = []
        for c in self._inflight:
            if c["ev"] is not None:
                main.wait_event(c["ev"])
            if not c["stepped"]:
                if "shard" in c:
                    self._step_shard(opt, c)
                else:
                    opt.launch(c["src"], c["b"], c["e"])
            if "shard" in c:
                regions.append((c["b"], c["e"]))
            covered += c["e"] - c["b"]
        if regions:  # region-wise sharded update: whose slices of the moments live where (gather_moments)
            opt._moments_sharded = self.world > 1 and any(self.shard_bounds(e - b, b)[0] > 0 for b, e in regions)
            opt._shard_regions = regions
        self._inflight.clear()
        if covered != opt.flat_g.numel():
            raise RuntimeError(f"staged gradient exchange covered {covered} of {opt.flat_g.numel()} elements")
        opt.step_count += 1

    # ---- sharded update: reduce-scatter -> Adam on 1 / world of the buffers -> all-gather ---------------------------------
    # An all-reduce IS a reduce-scatter followed by an all-gather; doing the optimizer step between the two halves moves the
    # same bytes over the links (the gradient sum in, the updated f32 parameters out) while every rank runs Adam over its own
    # 1 / world slice only: 0.75 GB of HBM traffic per step becomes 0.75 / world GB + one conversion pass that rebuilds the
    # bf16 operand copies from the gathered parameters (0.15 GB).  The moments of the other slices are never touched on
    # this rank (they stay zero): ``gather_moments`` -- a collective, called by the entry points on every rank before rank 0
    # saves -- rebuilds the full moment buffers for a checkpoint, and FlatAdam.state_dict() refuses to run without it.
    def shard_bounds(self, n: int, base: int = 0) -> tuple:
        """(slice length, begin, end of THIS rank's slice, end of the evenly sharded body) of the ``n`` elements that start at
        ``base`` (a region of the flat buffers; default: all of them).  Slices are multiples of 8 elements; [body, base + n) --
        fewer than 8 * world elements -- is all-reduced and stepped on every rank."""
        per = (n // self.world) // 8 * 8
        real = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        r = dist.get_rank(self.group) if real == self.world else 0  # (a dry run on fewer processes: the first slice)
        return per, base + r * per, base + (r + 1) * per, base + per * self.world

    # ---- region-wise sharded update (shard_update with the staged backward) ---------------------------------------------------------
    # Every region of the flat gradient is reduce-SCATTERED as soon as backward has finished it (the first half of the all-reduce it
    # replaces: the same bytes on the links, the same overlap with the rest of backward); each rank runs Adam over its 1 / world slice of
    # the region (0.75 GB of optimizer traffic per step and rank become 0.75 / world GB: the 130 us Adam pass at the end of the step is
    # ~16 us at 8 ranks), the updated f32 parameters are all-GATHERED (the all-reduce's second half), and one cast launch rebuilds the
    # bf16 operand copies of the slices other ranks stepped.  Elementwise the same update on the same summed gradient: parameters
    # equal to all-reduce + full Adam bit for bit (tests/test_dist_gloo.py); the moments of a slice live on the rank that steps it
    # (``gather_moments`` before a checkpoint).
    def _start_sharded(self, opt, lo, hi, after=()) -> None:
        flat_g = opt.flat_g
        gpu = flat_g.is_cuda
        compress = self.compress == "bf16" and gpu
        src = flat_g
        main = torch.cuda.current_stream(flat_g.device) if gpu else None
        if gpu and self._side is None:
            self._side = torch.cuda.Stream(device=flat_g.device)
        if not self._inflight and not self.hyper_ready:  # first region of this step: the step's Adam constants
            opt.grad_scale = 1.0 / self.world
            opt.prepare_hyper()
        for s_ in after:  # (weight-gradient side streams the compute stream has not waited for)
            (main if compress else self._side).wait_stream(s_)
        if compress:
            from . import _lib
            from .ops import _ck, _p, _stream
            if self._g16 is None or self._g16.numel() != flat_g.numel():
                self._g16 = torch.empty(flat_g.numel(), dtype=torch.bfloat16, device=flat_g.device)
            src = self._g16
            _ck(_lib.load().egk_cast(_stream(), _p(flat_g[lo:hi]), 0, _p(src[lo:hi]), 1, hi - lo), "egk_cast")
        per, a, b, body = self.shard_bounds(hi - lo, lo)
        real = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        native = gpu and real == self.world and dist.get_backend(self.group) == "nccl" and not _skip["collectives"]

        def scatter_sum():
            if per:
                if native:  # in place: this rank's slice of the region receives the sum
                    dist.reduce_scatter_tensor(src[a:b], src[lo:body], op=dist.ReduceOp.SUM, group=self.group)
                elif real == self.world:  # (gloo has no reduce-scatter: the sum of the whole body, of which one slice is used)
                    all_reduce_sum_(src[lo:body], self.group)
                else:
                    all_reduce_sum_(src[a:b], self.group)  # dry run: a collective of one slice on the group there is
            if body < hi:
                all_reduce_sum_(src[body:hi], self.group)
        ev = None
        if gpu:
            ready = torch.cuda.Event()
            ready.record(main)
            self._side.wait_event(ready)
            with torch.cuda.stream(self._side):
                scatter_sum()
                ev = torch.cuda.Event()
                ev.record(self._side)
        else:
            scatter_sum()
        self._inflight.append({"b": lo, "e": hi, "ev": ev, "src": src, "stepped": False, "shard": (per, a, b, body)})

    def _step_shard(self, opt, c) -> None:
        """On the current stream, behind the region's reduce-scatter: Adam over this rank's slice (+ the unsharded remainder), the
        all-gather of the updated parameters on the communication stream, the bf16 copies of the gathered slices."""
        per, a, b, body = c["shard"]
        lo, hi = c["b"], c["e"]
        gpu = opt.flat_p.is_cuda
        if per:
            opt.launch(c["src"], a, b)
        if body < hi:
            opt.launch(c["src"], body, hi)
        real = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if per and real == self.world and self.world > 1 and not _skip["collectives"]:
            flat_p = opt.flat_p

            def gather():
                if gpu and dist.get_backend(self.group) == "nccl":
                    dist.all_gather_into_tensor(flat_p[lo:body], flat_p[a:b], group=self.group)
                else:
                    mine = flat_p[a:b].cpu() if gpu else flat_p[a:b].clone()
                    parts = [torch.empty_like(mine) for _ in range(self.world)]
                    dist.all_gather(parts, mine, group=self.group)
                    for r, part in enumerate(parts):
                        flat_p[lo + r * per: lo + (r + 1) * per].copy_(part)
            if gpu:
                cur = torch.cuda.current_stream(flat_p.device)
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    gather()
                cur.wait_stream(self._side)
            else:
                gather()
        if per:
            opt.refresh_shadows(lo, body)  # the bf16 operand copies of the slices other ranks stepped (one cast over the region)

    # ---- sharded update of the WHOLE buffer in one piece (one-piece backward: reduce_and_step) -------------------------------------
    def _sharded_step(self, opt) -> None:
        flat_g, n = opt.flat_g, opt.flat_g.numel()
        per, lo, hi, body = self.shard_bounds(n)
        real = dist.get_world_size(self.group) if dist.is_initialized() else 1
        on_gpu = flat_g.is_cuda
        compress = self.compress == "bf16" and on_gpu
        src = flat_g
        if compress:
            from . import _lib
            from .ops import _ck, _p, _stream
            if self._g16 is None or self._g16.numel() != n:
                self._g16 = torch.empty(n, dtype=torch.bfloat16, device=flat_g.device)
            src = self._g16
            _ck(_lib.load().egk_cast(_stream(), _p(flat_g), 0, _p(src), 1, n), "egk_cast")
        native = on_gpu and real == self.world and dist.get_backend(self.group) == "nccl" and not _skip["collectives"]

        def scatter_sum():
            if per:
                if native:  # in place: this rank's slice of the buffer receives the sum
                    dist.reduce_scatter_tensor(src[lo:hi], src[:body], op=dist.ReduceOp.SUM, group=self.group)
                elif real == self.world:  # (gloo has no reduce-scatter: the sum of the whole body, of which one slice is used)
                    all_reduce_sum_(src[:body], self.group)
                else:
                    all_reduce_sum_(src[lo:hi], self.group)  # dry run: a collective of one slice on the group there is
            if body < n:
                all_reduce_sum_(src[body:n], self.group)

        def gather_params():
            if not per or _skip["collectives"]:
                return
            flat_p = opt.flat_p
            if native:
                dist.all_gather_into_tensor(flat_p[:body], flat_p[lo:hi], group=self.group)
            elif real == self.world:
                mine = flat_p[lo:hi].cpu() if on_gpu else flat_p[lo:hi].clone()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                for r, part in enumerate(parts):
                    flat_p[r * per:(r + 1) * per].copy_(part)
            # (a dry run on fewer processes has nothing to gather from)

        if on_gpu:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat_g.device)
            main = torch.cuda.current_stream(flat_g.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                scatter_sum()
            main.wait_stream(self._side)
        else:
            scatter_sum()
        opt.grad_scale = 1.0 / self.world
        opt.prepare_hyper()
        opt.launch(src, lo, hi)
        if body < n:
            opt.launch(src, body, n)
        if on_gpu:
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                gather_params()
            main.wait_stream(self._side)
        else:
            gather_params()
        opt.refresh_shadows()  # the bf16 operand copies of the slices other ranks stepped
        opt.step_count += 1
        opt._moments_sharded = self.world > 1 and per > 0  # (FlatAdam.state_dict refuses until gather_moments has run)
        opt._shard_regions = [(0, n)]

    def gather_moments(self, opt) -> None:
        """Sharded update: all-gather every rank's slice of the Adam moments so that each rank holds the full flat_m / flat_v
        -- what a checkpoint must contain (a resumed run may use another world size or the all-reduce path).  A COLLECTIVE:
        every rank calls it (the entry points do, before rank 0 saves).  The slices of the other ranks are overwritten by the
        next sharded step's bookkeeping only in the sense that this rank never reads them: the gathered values stay valid
        until the next step."""
        if not getattr(opt, "_moments_sharded", False):
            return
        if getattr(opt, "flat_m", None) is None:
            opt._moments_sharded = False
            return
        n = opt.flat_m.numel()
        regions = getattr(opt, "_shard_regions", None) or [(0, n)]
        if len(regions) > 1 or regions[0] != (0, n):
            return self._gather_moments_regions(opt, regions)
        per, lo, hi, body = self.shard_bounds(n)
        real = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if not per and real == self.world:
            opt._moments_sharded = False  # (fewer than 8 * world elements: the whole buffer is all-reduced and stepped on every rank)
            return
        if not (per and real == self.world):
            if real == 1 and self.world == 1:
                opt._moments_sharded = False
                return
            # nothing can be gathered (an empty shard, or the group's real size is not the world the update was sharded for: the
            # exchange dry run): the moments of the other ranks' slices are NOT in this process -- the flag stays set, so that
            # FlatAdam.state_dict() keeps refusing to serialise partial moments
            import warnings
            warnings.warn("GradSync.gather_moments: the sharded Adam moments cannot be gathered on this group "
                          f"(group size {real}, sharded for {self.world}); a checkpoint would hold partial moments")
            return
        native = opt.flat_m.is_cuda and dist.get_backend(self.group) == "nccl"
        for buf in (opt.flat_m, opt.flat_v):
            if native:
                dist.all_gather_into_tensor(buf[:body], buf[lo:hi].clone(), group=self.group)
            else:
                mine = buf[lo:hi].cpu() if buf.is_cuda else buf[lo:hi].clone()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                for r, part in enumerate(parts):
                    buf[r * per:(r + 1) * per].copy_(part)
        opt._moments_sharded = False

    def _gather_moments_regions(self, opt, regions) -> None:
        real = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if real != self.world:
            import warnings
            warnings.warn("GradSync.gather_moments: the sharded Adam moments cannot be gathered on this group "
                          f"(group size {real}, sharded for {self.world}); a checkpoint would hold partial moments")
            return
        native = opt.flat_m.is_cuda and dist.get_backend(self.group) == "nccl"
        for lo, hi in regions:
            per, a, b, body = self.shard_bounds(hi - lo, lo)
            if not per:
                continue
            for buf in (opt.flat_m, opt.flat_v):
                if native:
                    dist.all_gather_into_tensor(buf[lo:body], buf[a:b].clone(), group=self.group)
                else:
                    mine = buf[a:b].cpu() if buf.is_cuda else buf[a:b].clone()
                    parts = [torch.empty_like(mine) for _ in range(self.world)]
                    dist.all_gather(parts, mine, group=self.group)
                    for r, part in enumerate(parts):
                        buf[lo + r * per: lo + (r + 1) * per].copy_(part)
        opt._moments_sharded = False

    def reduce_and_step(self, opt) -> None:
        """Gradient exchange + optimizer step of a ``FlatAdam`` as a PIPELINE over chunks of the flat buffer: chunk i is
        converted (bf16 compression) on the compute stream, all-reduced on the communication stream, and stepped by
        its own Adam launch as soon as its collective is done -- the Adam launch of chunk i runs while chunks i+1..
        are still on the xGMI links (the one-shot form waited for the last collective before the first Adam byte)."""
        if self.shard_update:
            return self._sharded_step(opt)
        flat_g = opt.flat_g
        n = flat_g.numel()
        compress = self.compress == "bf16"
        src = flat_g
        if compress:
            from . import _lib
            from .ops import _ck, _p, _stream
            if self._g16 is None or self._g16.numel() != n:
                self._g16 = torch.empty(n, dtype=torch.bfloat16, device=flat_g.device)
            src = self._g16
        if self._side is None:
            self._side = torch.cuda.Stream(device=flat_g.device)
        main = torch.cuda.current_stream(flat_g.device)
        bounds = chunk_bounds(n, self.chunk_elems)
        done = []
        for b, e in bounds:
            if compress:
                _ck(_lib.load().egk_cast(_stream(), _p(flat_g[b:e]), 0, _p(src[b:e]), 1, e - b), "egk_cast")
            ready = torch.cuda.Event()
            ready.record(main)
            self._side.wait_event(ready)
            with torch.cuda.stream(self._side):
                all_reduce_sum_(src[b:e], self.group)
                ev = torch.cuda.Event()
                ev.record(self._side)
            done.append(ev)
        opt.grad_scale = 1.0 / self.world
        opt.prepare_hyper()
        for (b, e), ev in zip(bounds, done):
            main.wait_event(ev)
            opt.launch(src, b, e)
        opt.step_count += 1


def sync_parameters(optimizer, sync: GradSync):
    """Make every rank start from rank 0's parameters (after FlatAdam has materialised)."""
    sync.broadcast_(optimizer.flat_p)
