"""torch.autograd.Functions over the C ABI of libegopack_hip.so.

PyTorch is used for device memory, the current HIP stream and autograd bookkeeping only: every
piece of arithmetic below is a launch of a hand-written gfx950 kernel through ctypes.  There is
no CPU path: tensors must live on a ROCm device, otherwise a RuntimeError is raised.

Numeric modes (``set_compute``):
  'bf16'         bf16 MFMA (f32 accumulate); activations, activation gradients and the weight operands
                 (bf16 shadows written by the Adam kernel) are bf16 in HBM; statistics, losses, parameters,
                 parameter gradients and optimiser state are f32.  The benchmark mode.
  'bf16_f32act'  bf16 MFMA with f32 activations / weights in HBM (converted while staging).
  'f32'          exact f32 MFMA, everything f32: the tight-parity mode.
  'bf16x3'       f32 activations and statistics like 'f32', but every contraction runs on the bf16 matrix pipe as THREE products
                 of the operands' bf16 halves (x = hi + lo: a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, f32 accumulation; relative
                 error ~2^-17 per product instead of 2^-9): the forward-only, f32-grade feature path that feeds the
                 nearest-prototype index op in 'bf16' mode (``precise_features``), at ~3x the bf16 cost instead of the 16x of
                 the exact-f32 matrix instructions.
Every op is polymorphic in the activation element type: it follows the dtype of its activation input.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib, switches

F32, BF16 = 0, 1  # EGK_COMPUTE_* and EGK_F32 / EGK_BF16 element types
X3 = 2            # host-side pseudo compute type: f32 values contracted as three bf16 products per K source (``_x3_expand``)

_MODES = {"bf16": (BF16, torch.bfloat16), "bf16_f32act": (BF16, torch.float32), "f32": (F32, torch.float32),
          "bf16x3": (X3, torch.float32)}
_state = {"mode": "bf16", "compute": BF16, "act": torch.bfloat16}


def set_compute(mode: str) -> None:
    comp, act = _MODES[mode]
    _state.update(mode=mode, compute=comp, act=act)


def get_compute() -> str:
    return _state["mode"]


def act_dtype() -> torch.dtype:
    return _state["act"]


class compute_mode:
    """Context manager: ``with ops.compute_mode('f32'): ...``"""

    def __init__(self, mode):
        self.mode, self.prev = mode, None

    def __enter__(self):
        self.prev = get_compute()
        set_compute(self.mode)

    def __exit__(self, *a):
        set_compute(self.prev)


# ---- plumbing -----------------------------------------------------------------------------------
def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "egopack_amd: the HIP path needs tensors on a ROCm device (got a CPU tensor); there is "
                "no CPU fallback -- the CPU oracle under oracle/ is test infrastructure only")


def _materialise_virtual(t: torch.Tensor) -> None:
    """A lazily widened activation (``to_act(..., lazy=True)``: an f32 tensor whose bf16 source is known and whose own values
    were never written) is about to be READ as f32: write it now."""
    src = t._egk_bf16_src
    t._egk_virtual = False
    _ck(_lib.load().egk_cast(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(src.data_ptr()), BF16,
                             C.c_void_p(t.data_ptr()), F32, src.numel()), "egk_cast")


_slab_pending = []  # weak references to results whose reduce launch is still owed (``precise_scope`` settles them on its way out)


def _slab_mark(t: torch.Tensor, ws, M: int, N: int, bias) -> None:
    import weakref
    t._egk_slabs = (ws, M, N, bias)
    _slab_pending.append(weakref.ref(t))


def _slabs_settle() -> None:
    """Run the reduce launch of every result that still owes one (nobody slab-aware read it): nothing unreduced outlives the
    scope it was made in."""
    pending, _slab_pending[:] = list(_slab_pending), []
    for r in pending:
        t = r()
        if t is not None and getattr(t, "_egk_slabs", None) is not None:
            _materialise_slabs(t)


def _materialise_slabs(t: torch.Tensor) -> None:
    """The result of a split contraction whose reduce launch was left out (``gemm(defer_reduce=True)``: its two K slabs sit in a
    private workspace) is about to be read by something that does not take slabs: run the reduce launch now."""
    ws, M, N, bias = t._egk_slabs
    t._egk_slabs = None
    _ck(_lib.load().egk_gemm_reduce_slabs(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ws.data_ptr()), 2, M, N,
                                          C.c_void_p(bias.data_ptr()) if bias is not None else None, C.c_void_p(t.data_ptr()), N),
        "egk_gemm_reduce_slabs")
    _slabs_written(t)


def _slabs_written(t: torch.Tensor) -> None:
    """``t`` now holds the reduced matrix: a taped node that recorded it earlier (ops.dual_record) is ready from HERE on."""
    ent = getattr(t, "_egk_tape_entry", None)
    if ent is not None:
        ent["@event"] = torch.cuda.current_stream().record_event()
        t._egk_tape_entry = None


def _slab_consumer(x: torch.Tensor, cols_max: int = 1024):
    """(pointer to pass as the row kernel's input, True) after arming the one-shot slab input for ``x`` -- the f32 result of a
    split contraction whose reduce was deferred -- or (None, False): the caller goes on with ``x`` as it is (``_c`` / ``_p``
    materialise it).  The armed kernel reads slab 0 + slab 1 + bias and stores the reduced matrix into ``x``."""
    sl = getattr(x, "_egk_slabs", None)
    if sl is None:
        return None, False
    ws, M, N, bias = sl
    if not (x.is_contiguous() and x.dim() == 2 and tuple(x.shape) == (M, N) and N <= cols_max and N % 4 == 0):
        return None, False
    x._egk_slabs = None
    rc = _lib.load().egk_slab_input_next(C.c_void_p(ws.data_ptr() + 4 * M * N), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                         C.c_void_p(x.data_ptr()))
    if rc != 0:
        raise RuntimeError(f"egk_slab_input_next failed (code {rc}): {_lib.last_error()}")
    x._egk_slab_keep = (ws, bias)  # (alive until the tensor goes: the armed launch reads them)
    return C.c_void_p(ws.data_ptr()), True


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if getattr(t, "_egk_virtual", False):  # (any kernel that takes this tensor's pointer reads its f32 values)
        _materialise_virtual(t)
    if getattr(t, "_egk_slabs", None) is not None:
        _materialise_slabs(t)
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Launches forked to another stream are ISSUED one launch late: after the forking stream's NEXT library launch, behind an
# event recorded at the fork point.  Same dependencies, same work -- but under hipGraph capture the order in which the two
# children of a fork are created decides which of them stays on the parent's hardware queue (the first one), and the
# other pays a cross-queue hand-off.  A backward pass naturally issues the forked weight gradient first, so the dX CHAIN
# was the child that hopped, at every fork (tools/exp/fork_order.py: 12 links with a forked weight-gradient launch each,
# 601 us per replay side-launch-first, 500 us chain-first).
_deferred = {"items": [], "busy": False, "on": switches.enabled("fork_order")}


def set_deferred_forks(on: bool) -> bool:
    """Switch the late issue of forked launches (engine: per step class); returns the previous setting.  Anything still
    queued is dropped: a step ends with ``join_wgrad`` (which drains the queue), so something is left only when the step
    was abandoned by an exception -- its launches must not surface in the next step."""
    prev = _deferred["on"]
    _deferred["on"] = bool(on) and switches.enabled("fork_order")
    _deferred["items"] = []
    return prev


def defer_after_next_launch(fn) -> None:
    """Run ``fn(event)`` right after the current stream's next library launch (``event``: recorded on the current stream now);
    immediately if the mechanism is off.  ``fn`` issues work on OTHER streams behind ``event``."""
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    if not _deferred["on"]:
        fn(ev)
        return
    _deferred["items"].append(((cur.device.index, cur.cuda_stream), ev, fn))


def drain_deferred(all_streams: bool = True) -> None:
    """Issue the deferred side launches now (every join point calls this first)."""
    if _deferred["busy"] or not _deferred["items"]:
        return
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    take = [it for it in _deferred["items"] if all_streams or it[0] == key]
    if not take:
        return
    _deferred["items"] = [it for it in _deferred["items"] if not (all_streams or it[0] == key)]
    _deferred["busy"] = True
    try:
        for _, ev, fn in take:
            fn(ev)
    finally:
        _deferred["busy"] = False


def _ck(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {_lib.last_error()}")
    if _deferred["items"] and not _deferred["busy"]:
        drain_deferred(all_streams=False)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"expected a float32 or bfloat16 activation, got {t.dtype}")


def _c(t: torch.Tensor) -> torch.Tensor:
    """contiguous activation (f32 or bf16)"""
    _dt(t)
    if getattr(t, "_egk_virtual", False):
        _materialise_virtual(t)
    if getattr(t, "_egk_slabs", None) is not None:
        _materialise_slabs(t)
    return t if t.is_contiguous() else t.contiguous()


def _rm(t: torch.Tensor) -> torch.Tensor:
    """row-major 2-D matrix with unit column stride (any row stride: padded logits are views)"""
    _dt(t)
    if getattr(t, "_egk_virtual", False):
        _materialise_virtual(t)
    if getattr(t, "_egk_slabs", None) is not None:
        _materialise_slabs(t)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


_ws_cache = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream).  Every kernel that takes a workspace uses it
    only inside one API call, so sharing it between calls on one stream is safe."""
    key = (torch.device(device).index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_rng = {"seed": 0x5EED_E60, "offset": 0}


def manual_seed(seed: int) -> None:
    """Seed of the Philox dropout streams (per process; ranks should use different seeds)."""
    _rng["seed"], _rng["offset"] = int(seed) & 0xFFFFFFFFFFFFFFFF, 0
    for t in _rng_dev.values():  # the per-device replay offsets restart too: seeding twice gives the same masks twice
        t.zero_()


def _next_rng(n_elems: int):
    off = _rng["offset"]
    _rng["offset"] += (n_elems + 3) // 4 + 64
    return _rng["seed"], off


_rng_dev = {}


def rng_device_offset(device) -> torch.Tensor:
    """Device-resident uint64 added to every dropout launch's Philox offset.  Captured graphs bake the
    host-side (seed, offset) into their kernel arguments; advancing this word between replays
    (``advance_rng_device``) gives every replayed step fresh masks."""
    key = torch.device(device).index
    t = _rng_dev.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _rng_dev[key] = t
    return t


RNG_DEVICE_STRIDE = 1 << 40


def advance_rng_device(device, stride: int = RNG_DEVICE_STRIDE):
    rng_device_offset(device).add_(stride)


def rng_snapshot() -> int:
    """Host-side position of the dropout streams (see ``rng_replay``)."""
    return _rng["offset"]


class rng_replay:
    """``with ops.rng_replay(snap):`` -- the dropout launches issued inside draw the offsets the launches issued since
    ``snap = ops.rng_snapshot()`` drew: a second pass over the SAME sequence of dropout calls (same shapes) gets the same keep
    masks (the forward-only precise pass of the EgoPack step when the backbone runs in train mode).  The stream position
    afterwards is the furthest either pass reached."""

    def __init__(self, snap: int):
        self.snap = int(snap)

    def __enter__(self):
        self.resume = _rng["offset"]
        _rng["offset"] = self.snap

    def __exit__(self, *a):
        _rng["offset"] = max(self.resume, _rng["offset"])


def get_rng_state() -> dict:
    """State of the dropout streams (host seed / offset and the per-device replay offsets) for checkpoints."""
    return {"seed": _rng["seed"], "offset": _rng["offset"], "device": {k: int(v.item()) for k, v in _rng_dev.items()}}


def set_rng_state(state: dict) -> None:
    _rng["seed"], _rng["offset"] = int(state["seed"]), int(state["offset"])
    for k, v in state.get("device", {}).items():
        dev = torch.device("cuda", int(k)) if k is not None else torch.device("cuda")
        rng_device_offset(dev).fill_(int(v))


# ---- element-type conversion ---------------------------------------------------------------------------
def cast_raw(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Plain (non-differentiable) f32 <-> bf16 conversion on the HIP path."""
    if src.dtype == dtype:
        return src
    _need_gpu(src)
    src = _c(src)
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _ck(_lib.load().egk_cast(_stream(), _p(src), _dt(src), _p(dst), _dt(dst), src.numel()), "egk_cast")
    return dst


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return cast_raw(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return cast_raw(g, ctx.src), None


def to_act(x: torch.Tensor, lazy: bool = False) -> torch.Tensor:
    """Bring an activation to the element type of the current mode (no-op when it already has it).  ``lazy``: the caller hands
    the result straight to a contraction (the temporal pooling's first Linear).  In the three-product mode a bf16 input IS the
    high half of its widening and has no low half, so that contraction never reads the widened f32 values: the f32 tensor is then
    only allocated, and written when anything else asks for its pointer (``_p`` / ``_c`` / ``_rm``) -- 15 us and 38 MB at the head
    of the precise pass of BASELINE config 4 (EGK_DISABLE=x3_lazy_input: always written)."""
    want = _state["act"]
    if x.dtype == want:
        return x
    if x.requires_grad:
        return _Cast.apply(x, want)
    if (lazy and _state["compute"] == X3 and want == torch.float32 and x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_cuda
            and x.is_contiguous() and not torch.is_grad_enabled() and switches.enabled("x3_lazy_input")):
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        y._egk_bf16_src, y._egk_virtual = x, True
        return y
    y = cast_raw(x, want)
    if _state["compute"] == X3 and x.dtype == torch.bfloat16:
        y._egk_bf16_src = x if x.is_contiguous() else x.contiguous()  # (its lo half is zero: one product less, no split pass)
    return y


def weight_operand(W: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The tensor a contraction reads for parameter ``W``: W itself (f32) or its bf16 shadow.  Shadows are
    owned and kept fresh by optim.FlatAdam (written by the Adam kernel); without one the weight is
    converted on the fly."""
    if dtype == torch.float32:
        return _f32c(W)
    sh = getattr(W, "_egk_shadow", None)
    if sh is not None and sh.shape == W.shape:
        return sh
    if not W.requires_grad:
        # a FROZEN weight (the auxiliary tasks' heads of the EgoPack step: engine.EgoPackStep freezes the tasks it does not
        # train): its bf16 copy is made once and kept while the parameter is not written (tensor version counter) -- it used to
        # be converted at every use, six launches per EgoPack step
        c = getattr(W, "_egk_frozen_copy", None)
        if c is not None and c[0] == W._version and c[1] == W.data_ptr() and c[2].shape == W.shape:
            return c[2]
        if not torch.cuda.is_current_stream_capturing():
            copy = cast_raw(W.detach(), torch.bfloat16)
            try:
                W._egk_frozen_copy = (W._version, W.data_ptr(), copy)
            except Exception:  # noqa: BLE001  (a tensor type that takes no attributes)
                pass
            return copy
    return cast_raw(W.detach(), torch.bfloat16)


# ---- raw GEMM ------------------------------------------------------------------------------------
def _gemm_desc(M, N, A1, lda1, B1, ldb1, K1, out, ldc, *, A2=None, lda2=0, B2=None, ldb2=0, K2=0, transA=False,
               transB=False, bias=None, residual=None, ldr=0, act=0, accumulate=False, alpha=1.0, compute=None,
               dbias=None, into=None, stats=None, op_f16=False, slot_final=True):
    """Fill an ``egk_gemm_desc`` (a fresh one, or ``into``: an element of a descriptor array).  ``stats``: per-segment sums of
    the result for the graph LayerNorm that consumes it, taken in the epilogue (``_ln_stats_request``)."""
    op_dt = _dt(A1)
    compute = (BF16 if op_dt == BF16 else _state["compute"]) if compute is None else compute
    extra = []
    if compute == X3:
        srcs = _x3_expand(M, N, [(A1, lda1, B1, ldb1, K1)] + ([(A2, lda2, B2, ldb2, K2)] if A2 is not None and K2 else []),
                          transA, transB)
        if srcs is None:  # (shapes the pipelined kernel does not take: the exact-f32 matrix instructions instead)
            compute = F32
        else:
            if dbias is not None:
                # the kernel's fused bias gradient sums the A image of a SINGLE K source: with three products in the launch
                # the column sum of dY (A as stored: [K rows, M columns]) is a launch of its own, in f32
                _colsum_into(A1.as_strided((K1, M), (lda1, 1)), dbias, True)
                dbias = None
            compute, op_dt = BF16, BF16
            (A1, lda1, B1, ldb1, K1), rest = srcs[0], srcs[1:]
            A2 = B2 = None
            lda2 = ldb2 = K2 = 0
            if rest:
                (A2, lda2, B2, ldb2, K2), extra = rest[0], rest[1:]
    if _dt(A1) != op_dt or _dt(B1) != op_dt or (A2 is not None and (_dt(A2) != op_dt or _dt(B2) != op_dt)):
        raise TypeError("gemm: all A / B operands must share one element type")
    d = _lib.GemmDesc() if into is None else into
    d.M, d.N, d.K1, d.K2 = M, N, K1, K2
    d.A1, d.A2, d.B1, d.B2 = _p(A1), _p(A2), _p(B1), _p(B2)
    d.lda1, d.lda2, d.ldb1, d.ldb2 = lda1, lda2, ldb1, ldb2
    d.transA, d.transB = int(transA), int(transB)
    d.a_dtype = d.b_dtype = op_dt
    d.n_extra = len(extra)
    for i, (xa, xlda, xb, xldb, xk) in enumerate(extra):
        d.xA[i], d.xB[i], d.xlda[i], d.xldb[i], d.xK[i] = xa.data_ptr(), xb.data_ptr(), xlda, xldb, xk
    d.c_dtype = _dt(out)
    d.op_f16 = 1 if op_f16 else 0  # (grouped launches: the 16-bit operands hold IEEE half values, see nearest_prototypes_grouped)
    d.compute = compute
    d.C, d.ldc = _p(out), ldc
    d.accumulate, d.act, d.alpha = int(accumulate), act, alpha
    d.bias, d.residual, d.ldr = _p(bias), _p(residual), ldr
    d.r_dtype = _dt(residual) if residual is not None else F32
    d.splitk = 1
    d.dbias = _p(dbias)
    d.ws, d.ws_bytes = None, 0
    # A dW-form launch that accumulates into a parameter's slot of the flat gradient buffer STORES instead when the step installed a
    # provider that says this launch is the slot's only writer in this step (engine.StepBase._grad_store_begin, optim.FlatAdam.store_begin);
    # ``slot_final`` False: several launches add up this gradient -- the provider is not asked
    prov = _slot_provider["provider"]
    if (prov is not None and slot_final and transA and transB and accumulate and compute == BF16 and bias is None and residual is None and act == 0
            and alpha == 1.0 and stats is None and out.dtype == torch.float32):
        if prov(out, M, N, ldc) == "store":
            d.accumulate = 0
    d.st_mode = 0
    if stats is not None:
        d.st_mode, d.st_nseg, d.st_min_seg_rows = stats["mode"], stats["n_seg"], stats["min_rows"]
        d.st_seg_ptr, d.st_ws = _p(stats["seg_ptr"]), _p(stats.get("ws"))
        if stats["mode"] == 2:
            x = stats["x"]
            d.st_x, d.st_ldx, d.st_stats = _p(x), x.stride(0), _p(stats["stats"])
            d.st_w, d.st_b, d.st_slope = _p(stats["w"]), _p(stats["b"]), stats["slope"]
    return d


def _desc_k(d) -> int:
    return d.K1 + d.K2 + sum(d.xK[i] for i in range(d.n_extra))


# ---- three-product contraction of f32 values on the bf16 matrix pipe ('bf16x3') -------------------------------------------
# x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (egk_split_bf16).  a.b is taken as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi -- three
# K sources of ONE launch of the pipelined kernel (egk_gemm_desc extra sources), f32 accumulation; the dropped a_lo.b_lo term
# is ~2^-18 relative.  Where the halves come from:
#   * a weight in the optimizer's flat buffers: hi IS its bf16 shadow (written by the Adam kernel), lo comes from the flat lo
#     buffer (optim.FlatAdam.refresh_lo_shadows: one launch per region and step);
#   * any other weight / the prototype banks: split once and kept while (version, address) stand;
#   * an activation: split when first used and kept for the duration of a ``precise_scope`` (the same activation feeds
#     several contractions); an f32 activation that ``to_act`` widened from bf16 has lo = 0 and needs ONE product less.
_x3 = {"cache": None, "keep": []}


class precise_scope:
    """``with ops.precise_scope():`` -- compute mode 'bf16x3' inside, with the activation splits shared between the contractions
    that read the same tensor."""

    def __enter__(self):
        self.prev_mode, self.prev_cache = get_compute(), _x3["cache"]
        set_compute("bf16x3")
        _x3["cache"] = {}
        return self

    def __exit__(self, *a):
        _slabs_settle()
        set_compute(self.prev_mode)
        _x3["cache"] = self.prev_cache


def _split_rows(x, rows, cols, ld, want_hi=True):
    lib = _lib.load()
    hi = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device) if want_hi else None
    lo = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    _ck(lib.egk_split_bf16(_stream(), _p(x), ld, _p(hi), _p(lo), cols, rows, cols), "egk_split_bf16")
    return hi, lo


def _x3_act(A, rows, cols, ld):
    """(hi, lo or None, ld of the halves) of an f32 operand stored as [rows, cols] with row stride ``ld`` (row-major A: [M, K];
    a k-major one: [K, M])."""
    src = getattr(A, "_egk_bf16_src", None)
    if src is not None and src.shape == A.shape and A.dim() == 2 and tuple(A.shape) == (rows, cols) and ld == cols:
        return src, None, cols  # widened from bf16 by to_act: the value IS its bf16 half
    key = (A.data_ptr(), rows, cols, ld, A._version)
    cache = _x3["cache"]
    if cache is None:
        # outside a precise_scope (the mode as a TRAINING mode): the halves ride on the tensor itself, so the activation a
        # layer split in forward is not split again for its weight gradient, and a gradient is split once for dX and dW
        own = getattr(A, "_egk_x3", None)
        if own is not None and own[0] == key:
            return own[1], own[2], cols
        hi, lo = _split_rows(A, rows, cols, ld)
        try:
            A._egk_x3 = (key, hi, lo)
        except Exception:  # noqa: BLE001
            _x3["keep"].append((hi, lo, A))
        return hi, lo, cols
    hit = cache.get(key)
    if hit is None:
        hi, lo = _split_rows(A, rows, cols, ld)
        hit = cache[key] = (hi, lo, A)  # (A is held so that its address cannot be handed to another tensor while the entry lives)
    return hit[0], hit[1], cols


def _x3_weight(B, rows, cols, ld):
    """(hi, lo) of the B operand stored as [rows, cols] ([N, K], or [K, N] for a k-major one): a weight in the optimizer's flat
    buffers brings its halves along, a frozen one is split once, anything else (an activation standing as B in a weight
    gradient) is split here."""
    sh, lo = getattr(B, "_egk_shadow", None), getattr(B, "_egk_lo", None)
    if sh is not None and lo is not None and tuple(sh.shape) == (rows, cols) and ld == cols:
        lo_view, fresh, refresh = lo
        if not fresh():
            refresh()
        return sh, lo_view
    c = getattr(B, "_egk_split", None)
    if c is not None and c[0] == B._version and c[1] == B.data_ptr() and tuple(c[2].shape) == (rows, cols):
        return c[2], c[3]
    if sh is not None and lo is None and not torch.cuda.is_current_stream_capturing():
        init = getattr(B, "_egk_lo_init", None)  # in the flat buffers, low halves not allocated yet (optim.FlatAdam)
        if init is not None and init() and getattr(B, "_egk_lo", None) is not None:
            return _x3_weight(B, rows, cols, ld)
    if sh is None and getattr(B, "_egk_bf16_src", None) is not None:
        hi, lo_, _ = _x3_act(B, rows, cols, ld)  # an activation as B: the activation rules (a widened one has no low half)
        return hi, lo_
    hi, lo = _split_rows(B, rows, cols, ld)
    if sh is None and not B.requires_grad and not torch.cuda.is_current_stream_capturing():
        try:  # (frozen weights and banks: as weight_operand's copies; an in-place update moves ``_version``)
            B._egk_split = (B._version, B.data_ptr(), hi, lo)
        except Exception:  # noqa: BLE001
            _x3["keep"].append((hi, lo))
    else:
        _x3["keep"].append((hi, lo))
    return hi, lo


def _x3_expand(M, N, logical, transA, transB):
    """[(A, lda, B, ldb, K)] f32 sources -> the bf16 sources of the three-product contraction, or None when the pipelined kernel
    cannot take them (a K that is not a multiple of 64, rows that are not 16-byte aligned as bf16, more than two sources).
    Operands keep their layout: a k-major (transposed) operand is split as the [K, rows] array it is stored as."""
    if len(logical) > 2:
        return None
    shapes = []
    for A, lda, B, ldb, K in logical:
        ar, ac = (K, M) if transA else (M, K)
        br, bc = (K, N) if transB else (N, K)
        if (K % 64 or K == 0 or A.dtype != torch.float32 or B.dtype != torch.float32 or lda % 4 or ldb % 4 or ac % 8 or bc % 8
                or A.data_ptr() % 16 or B.data_ptr() % 16):
            return None
        shapes.append((ar, ac, br, bc))
    out = []
    for (A, lda, B, ldb, K), (ar, ac, br, bc) in zip(logical, shapes):
        ah, al, a_ld = _x3_act(A, ar, ac, lda)
        bh, bl = _x3_weight(B, br, bc, ldb)
        out.append((ah, a_ld, bh, bc, K))
        if bl is not None:
            out.append((ah, a_ld, bl, bc, K))
        if al is not None:
            out.append((al, ac, bh, bc, K))
    return out


def _x3_release():
    if _x3["keep"]:
        _x3["keep"] = []


def _tee_arm(y: torch.Tensor):
    """Inside a ``precise_scope``: arm the library's one-shot split tee (egk_tee_split_next) so that the row kernel launched
    NEXT -- the producer of the f32 [rows, cols] activation ``y`` -- also stores y's bf16 halves: the contraction that reads y
    then needs no egk_split_bf16 launch of its own (12 such launches sat on the precise pass's chain in BASELINE config 4).
    -> the halves, for ``_tee_done`` right after the launch, or None (not in a scope, not an f32 matrix, switched off)."""
    if (_x3["cache"] is None or _state["compute"] != X3 or y.dtype != torch.float32 or y.dim() != 2 or not y.is_contiguous()
            or y.numel() == 0 or not switches.enabled("x3_tee")):
        return None
    rows, cols = y.shape
    hi = torch.empty((rows, cols), dtype=torch.bfloat16, device=y.device)
    lo = torch.empty((rows, cols), dtype=torch.bfloat16, device=y.device)
    rc = _lib.load().egk_tee_split_next(_p(hi), _p(lo), cols)  # (not through _ck: nothing may be launched in between)
    if rc != 0:
        raise RuntimeError(f"egk_tee_split_next failed (code {rc}): {_lib.last_error()}")
    return hi, lo


def _tee_done(y: torch.Tensor, halves) -> None:
    if halves is not None:
        rows, cols = y.shape
        _x3["cache"][(y.data_ptr(), rows, cols, cols, y._version)] = (halves[0], halves[1], y)


def _gemm_deferrable(args, kw, slab_ok: bool = True) -> None:
    """``gemm(*args, **kw)`` of the precise pass (no gradient, inside a precise_scope, a plain f32 result with at most a bias): a
    launch that splits K in two leaves its slabs to the row kernel that reads the result next (``out._egk_slabs``, ``_slab_consumer``)
    -- when the caller knows that such a kernel is what follows (``slab_ok``)."""
    M, N, out = args[0], args[1], args[7]
    defer = (slab_ok and kw.get("compute") == X3 and _slab_defer["on"] and _x3["cache"] is not None and not torch.is_grad_enabled()
             and not kw.get("act") and kw.get("residual") is None and not kw.get("accumulate") and out.dtype == torch.float32
             and out.is_contiguous() and args[8] == N and N <= 1024 and N % 4 == 0)
    ws = gemm(*args, defer_reduce=defer, **kw)
    if ws is not None:
        _slab_mark(out, ws, M, N, kw.get("bias"))


def _gemm_with_stats(args, kw, stats):
    """``gemm(*args, **kw)`` with the epilogue statistics ``stats`` if the tile variant of this launch can take them:
    returns (partials, blocks) or None (plain launch done instead).  No split-K (the statistics need the finished tile)."""
    lib = _lib.load()
    d = _gemm_desc(*args, stats=stats, **kw)
    blocks = lib.egk_gemm_stats_blocks(C.byref(d))
    if blocks > 0 and d.n_extra and switches.enabled("x3_stats_split"):
        # a three-product contraction (six K sources for a SAGE layer's two-source launch: K = 6144 at H = 1024) of a batch
        # that fills half the chip: the statistics epilogue needs the finished tile, i.e. NO split-K -- 87 us for 2048 x 1024
        # on one workgroup per CU.  When the policy would cut the walk, the cut launch + its reduce + the LayerNorm's own
        # statistics pass are the cheaper chain (the precise pass of the EgoPack step sits on the step's critical path):
        # config 4 3.70 -> 3.65 ms.  ON BY DEFAULT since late round 4 (EGK_DISABLE=x3_stats_split turns it off; DESIGN 10.8,
        # profiles/r04_switches.txt): it was opt-in until the block-wise parity test of the OSCC head stated its input gradient "up
        # to near-ties of the max pool" -- the other summation order moves the auxiliary features by 1e-6 and with them WHICH one to
        # three near-tie pairs the draw contains (1.7e-3 .. 6.5e-3 over all rows, 1.66e-3 over the rows no tie touches).
        if lib.egk_gemm_splitk(d.M, d.N, _desc_k(d), d.compute) > 1:
            blocks = 0
    if blocks <= 0:
        _gemm_deferrable(args, kw)
        return None
    ws = torch.empty(blocks * stats["n_seg"] * 2, dtype=torch.float64, device=args[7].device)
    d.st_ws = _p(ws)
    _ck(lib.egk_gemm(_stream(), C.byref(d)), "egk_gemm")
    _x3_release()
    return ws, blocks


_slot_provider = {"provider": None}


def set_grad_slot_provider(provider):
    """``provider(out, M, N, ldc) -> "store" | None`` for dW-form launches whose result ``out`` is a parameter's gradient slot
    (optim.FlatAdam.learn_begin: counts the launches per slot during an eager step; .store_begin: answers "store" for the slots
    learnt as written once per step).  None switches it off.  Returns the previous provider."""
    prev = _slot_provider["provider"]
    _slot_provider["provider"] = provider
    return prev


def gemm(M, N, A1, lda1, B1, ldb1, K1, out, ldc, *, allow_splitk=True, splitk=None, defer_reduce=False, **kw):
    """``defer_reduce``: if the launch splits K in two, leave the slabs in a workspace of their own and do not launch the reduce
    (egk_gemm_defer_reduce_next) -- returns that workspace ([2][M][N] f32; ``out`` is NOT written, the bias NOT applied), else None."""
    lib = _lib.load()
    d = _gemm_desc(M, N, A1, lda1, B1, ldb1, K1, out, ldc, **kw)
    sk = lib.egk_gemm_splitk(M, N, _desc_k(d), d.compute) if allow_splitk else 1
    d.splitk = sk if splitk is None else int(splitk)
    need = lib.egk_gemm_ws_bytes(C.byref(d))
    deferred = None
    if need:
        if (defer_reduce and d.splitk == 2 and d.compute == BF16 and d.c_dtype == F32 and ldc == N
                and need == 8 * M * N):
            ws = deferred = torch.empty(need, dtype=torch.uint8, device=out.device)
        else:
            ws = workspace(need, out.device)
        d.ws, d.ws_bytes = _p(ws), ws.numel()
    if deferred is not None:
        lib.egk_gemm_defer_reduce_next(1)
    try:
        _ck(lib.egk_gemm(_stream(), C.byref(d)), "egk_gemm")
    finally:
        if deferred is not None:
            lib.egk_gemm_defer_reduce_next(0)
        _x3_release()
    return deferred


def gemm_grouped(problems, four_wave: bool = False):
    """ONE launch for up to 8 independent contractions of the same layout (``problems``: a list of (args, kwargs) of
    ``gemm`` without the split-K options; bf16 operands, K sources multiples of 64)."""
    lib = _lib.load()
    arr = (_lib.GemmDesc * len(problems))()
    for i, (a, kw) in enumerate(problems):
        _gemm_desc(*a, into=arr[i], **kw)
    if switches.debug("group_shapes") and not torch.cuda.is_current_stream_capturing():
        lay = ("t" if problems[0][1].get("transA") else "n") + ("t" if problems[0][1].get("transB") else "n")
        print(f"[group_shapes] {lay} x{len(problems)}: " + ", ".join(f"{a[0]}x{a[1]}x{a[6]}" for a, _ in problems)
              + f"  ({sum(((a[0] + 127) // 128) * ((a[1] + 127) // 128) for a, _ in problems)} tiles of 128 x 128)", flush=True)
    prev = None
    if four_wave:  # launch-shape hint: the 4-wave 128 x 128 variant whatever the tile count (the two-wave-group variant it
        prev = lib.egk_gemm_set_pipeline(3)  # replaces splits K over its two groups: same sums in another order)
        if prev != 1:  # (a development setting is in force: leave it)
            lib.egk_gemm_set_pipeline(prev)
            prev = None
    try:
        _ck(lib.egk_gemm_grouped(_stream(), arr, len(problems)), "egk_gemm_grouped")
    finally:
        _x3_release()
        if prev is not None:
            lib.egk_gemm_set_pipeline(prev)


def _colsum_into(x2d: torch.Tensor, out: torch.Tensor, accumulate: bool):
    lib = _lib.load()
    M, N = x2d.shape
    ws = workspace(lib.egk_colsum_ws_len(M, N) * 4, x2d.device)
    _ck(lib.egk_colsum(_stream(), _p(x2d), x2d.stride(0), M, N, _p(out), int(accumulate), _p(ws), _dt(x2d)), "egk_colsum")


def _grad_slot(param: Optional[torch.Tensor]):
    """Fused weight-gradient accumulation: if the parameter already owns a contiguous f32 .grad (the
    flat gradient buffer of egopack_amd.optim.FlatAdam), backward kernels accumulate into it and
    autograd gets None for that input."""
    if param is None:
        return None
    g = getattr(param, "grad", None)
    if g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.is_cuda:
        return g
    return None


# ---- weight-gradient side streams --------------------------------------------------------------------------------
# A dW contraction (64 tiles of 128 x 128, walked as <= 256 workgroups) feeds nothing but the optimizer, while the
# dX chain it would sit in is a strict dependency chain of launches, many of them small.  When the gradient lands in
# the flat buffer (no tensor goes back to autograd) the launch is put on a side stream that forks from the
# backward stream at that point; ``join_wgrad()`` (queued as an end-of-backward callback of the autograd engine) joins
# every side stream with the stream that called backward() before the gradients can be consumed.
# Off by default: measured on MI355X it gains 3-4 % on the multi-task steps (M = 6144 backbone rows, heads on their
# own streams) and loses 3-5 % on single-task steps (M = 2048: every launch is already a partial-chip launch);
# egopack_amd.engine turns it on for steps with more than one enabled task.
_wgrad = {"enabled": False, "streams": {}, "pending": [], "queued": False, "exclude": set()}


def exclude_wgrad_streams(streams) -> None:
    """Backward work running on these streams keeps its weight-gradient launches (the task-head streams of the
    engine already overlap each other; forking each of them again oversubscribes the hardware queues)."""
    for st in streams:
        _wgrad["exclude"].add((st.device.index, st.cuda_stream))



def scope_excluded_streams(streams) -> None:
    """The excluded set becomes exactly ``streams`` (a step calls this with ITS head / task / side streams when it starts issuing:
    stream handles are pooled and handed out round-robin, so in a process that builds many steps -- a test suite -- the handles
    excluded by steps long gone would otherwise pile up until every stream of the pool, the backward stream of the next step
    included, counts as a head stream and its weight gradients take other launch paths than the same step in a fresh process)."""
    _wgrad["exclude"] = {(st.device.index, st.cuda_stream) for st in streams if st is not None}


def unexcluded_stream() -> torch.cuda.Stream:
    """A stream from the pool whose handle is not registered as a head / task stream (torch hands pooled handles out
    round-robin: a fresh ``torch.cuda.Stream()`` may alias one an earlier step excluded) -- for graph captures."""
    keep = []
    for _ in range(64):
        st = torch.cuda.Stream()
        if (st.device.index, st.cuda_stream) not in _wgrad["exclude"]:
            return st
        keep.append(st)
    return keep[-1]


def set_wgrad_side_streams(on: bool) -> bool:
    prev = _wgrad["enabled"]
    _wgrad["enabled"] = bool(on)
    return prev


def _wgrad_launch(in_place: bool, tensors, launch, in_backward: bool = True):
    """Run ``launch()`` (weight / bias gradient kernels that only write persistent gradient slots) on the side
    stream of the current stream; ``tensors`` are the temporaries it reads (kept alive for that stream).
    ``in_backward`` False: called from ``join_wgrad`` itself (possibly after backward has returned): no end-of-backward
    callback is installed, the caller joins right away."""
    if not (_wgrad["enabled"] and in_place and tensors and tensors[0].is_cuda) or _wq_sched["mode"] == "inline":
        launch()
        return
    main = torch.cuda.current_stream()
    key = (main.device.index, main.cuda_stream)
    if key in _wgrad["exclude"]:
        launch()
        return
    side = _wgrad["streams"].get(key)
    if side is None:
        side = _wgrad["streams"][key] = torch.cuda.Stream(device=main.device)

    def issue(ev):
        side.wait_event(ev)
        with torch.cuda.stream(side):
            launch()
        for t in tensors:
            if t is not None:
                t.record_stream(side)
    defer_after_next_launch(issue)  # (the closure keeps the operands alive until then)
    if side not in _wgrad["pending"]:
        _wgrad["pending"].append(side)
    if in_backward and not _wgrad["queued"]:  # join at the end of THIS backward pass, on the stream of the thread that called it
        _wgrad["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(join_wgrad)


# ---- deferred, grouped weight gradients ------------------------------------------------------------------------------------
# A weight gradient of an H x H layer is a 64-tile contraction over K = all nodes: alone it fills a quarter of the chip,
# and split-K to fill it costs slabs plus a second launch (measured: 23 us + 8.7 us per launch, 12 of them in the headline
# step, all competing with the dX chain for the CUs).  With the queue on, such launches are PARKED (operands kept alive)
# and issued SIX AT A TIME as one grouped launch on the side stream: 384 workgroups, no slabs, no reduce launch, a sixth
# of the forks.  ``flush_wgrad`` issues what is parked (fewer than four at the end of backward).
_wq = {"on": False, "items": [], "tiles": 0, "hold": [], "extra": [], "riders": []}
# Six per launch: alone, six H x H problems in one launch of two 4-wave workgroups per CU run at 900 TF/s against 740 for
# four on one 8-wave workgroup per CU (tools/gemm_group_bench.py: x4 70 us, x6 85 us, x8 133 us at K = 6144).  Inside the
# step: 4 / 5 / 6 / 7 / 8 -> 1.558 / 1.60 / 1.538 / 1.60 / 1.595 ms (three alternating rounds of 200 steps; 12 H x H weight
# gradients = two full launches of six).  Before the dX chain kept its hardware queue at the forks (defer_after_next_launch)
# six measured WORSE than four (1.75 vs 1.68): the chain was the child that paid the queue hand-off behind every larger
# launch.  EGK_WGRAD_COUNT is a development knob (the library takes up to 8 per launch).
WGRAD_GROUP_COUNT = int(os.environ.get("EGK_WGRAD_COUNT", "6"))
WGRAD_GROUP_TILES = 64 * WGRAD_GROUP_COUNT
F32_WGRAD_GROUP_COUNT = int(os.environ.get("EGK_F32_WGRAD_COUNT", "8"))  # (development knob, as EGK_WGRAD_COUNT)


# WHEN parked weight gradients are issued (development knob EGK_WGRAD_SCHED, A/B of the backward schedule):
#   free   (default) a full group goes to the side stream as soon as it is full and runs beside whatever the dX chain does next
#   rows   what is parked is issued right before a backward ROW kernel (graph / row LayerNorm backward) and JOINED behind it:
#          the matrix-bound weight gradients then overlap the memory-bound row kernels only, never the chain's contractions
#          (two matrix-bound launches side by side each slow down by more than they overlap: VERDICT r3 timeline, a 38 us dX
#          contraction took 132 us beside a 184 us grouped weight-gradient launch)
#   inline every group is issued on the backward stream itself (no side stream at all)
# Measured (round 4, headline step, same box, three alternating rounds): free 1.474-1.477 ms, inline 1.598-1.604, rows
# 1.774-1.782 -- letting the weight gradients run beside WHATEVER the chain does is worth 125 us against serialising them, and
# a join behind every row kernel stalls the chain for the length of each group.
_wq_sched = {"mode": os.environ.get("EGK_WGRAD_SCHED", "free")}


def _rows_fork() -> bool:
    """'rows' schedule: called right before a backward row kernel is launched on the backward stream."""
    if _wq_sched["mode"] != "rows" or not (_wq["on"] and _wgrad["enabled"]) or not (_wq["items"] or _wq["extra"]) or _on_excluded_stream():
        return False
    flush_wgrad()  # (issued behind the row kernel's own launch: defer_after_next_launch)
    return True


def _rows_join(forked: bool) -> None:
    if not forked:
        return
    drain_deferred(all_streams=False)
    main = torch.cuda.current_stream()
    side = wgrad_side_stream(main)
    if side is not None:
        main.wait_stream(side)


def set_wgrad_grouping(on, count: Optional[int] = None):
    """Switch the parking queue (engine: at the start and the end of every step); returns the previous setting (hand it back to
    restore).  ``count``: bf16 problems per grouped launch for this step (None: WGRAD_GROUP_COUNT; EGK_WGRAD_COUNT overrides).
    Whatever is still parked is dropped: a step ends with ``join_wgrad`` (which issues it), so something is left only when the step
    was abandoned by an exception -- its weight gradients must not be accumulated by the next step."""
    prev = (_wq["on"], _wq.get("count"))
    if isinstance(on, tuple):
        on, count = on
    _wq["on"] = bool(on)
    _wq["count"] = None if (count is None or "EGK_WGRAD_COUNT" in os.environ) else int(count)
    _wq["items"], _wq["hold"], _wq["extra"], _wq["tiles"], _wq["riders"] = [], [], [], 0, []
    return prev


def park_rider(fn, hold=()) -> None:
    """``fn()`` -- a launch that feeds nothing on the backward chain (a reported vector, say) -- rides with the next flush of the
    parked weight gradients: issued on their side stream, in front of the grouped launch.  Run at once when nothing can be parked
    (queue or side streams off, or on a task-head stream, whose parked work is issued by another stream)."""
    if not (_wq["on"] and _wgrad["enabled"]) or _on_excluded_stream():
        fn()
        return
    _wq.setdefault("riders", []).append(fn)
    _wq["hold"].extend(t for t in hold if t is not None)


def _wgrad_groupable(M, N, A, lda, B, ldb, K, compute=None) -> bool:
    if not (_wq["on"] and _wgrad["enabled"]) or A.dtype != B.dtype:
        return False
    if A.dtype == torch.bfloat16:
        ok = K % 64 == 0 and lda % 8 == 0 and ldb % 8 == 0
    elif A.dtype == torch.float32 and compute == F32 and switches.enabled("f32_wgrad_groups"):
        ok = K % 32 == 0 and lda % 4 == 0 and ldb % 4 == 0  # exact-f32 problems: the grouped f32 kernel (egk_gemm_grouped)
    else:
        return False
    return (ok and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0 and ((M + 127) // 128) * ((N + 127) // 128) <= 128)


def _on_excluded_stream() -> bool:
    cur = torch.cuda.current_stream()
    return (cur.device.index, cur.cuda_stream) in _wgrad["exclude"]


def _wgrad_defer(args, kw, tensors, park_on_excluded: bool = False, park_only: bool = False) -> bool:
    """Park the dW contraction ``gemm(*args, **kw)`` (transA, transB, accumulate into a gradient slot) for the next grouped
    launch.  False: not eligible (the caller launches it itself).  On an excluded stream (the task-head streams of the
    engine) nothing is parked unless ``park_on_excluded``: such a problem is only PARKED there -- it is issued by a later
    flush from the backward stream, which by then has joined the head streams (engine._run_heads), never from the head
    stream itself, where the operands parked by the OTHER head streams would be raced."""
    M, N, A, lda, B, ldb, K = args[:7]
    excluded = _on_excluded_stream()
    if not _wgrad_groupable(M, N, A, lda, B, ldb, K, kw.get("compute")) or (excluded and not park_on_excluded):
        return False
    _wq["items"].append((args, kw))
    _wq["hold"].extend(t for t in tensors if t is not None)
    _wq["tiles"] += ((M + 127) // 128) * ((N + 127) // 128)
    if excluded or park_only:  # (park_only: a later weight-gradient launch of the same step takes it along, see _take_parked)
        return True
    # exact-f32 problems are matrix-pipe bound: a launch lasts as long as the workgroups on its fullest CU, so EIGHT H x H
    # problems (512 tiles = two per CU everywhere) where the bf16 launches take six
    count = F32_WGRAD_GROUP_COUNT if A.dtype == torch.float32 else (_wq.get("count") or WGRAD_GROUP_COUNT)
    if len(_wq["items"]) >= count or _wq["tiles"] >= 64 * count:
        flush_wgrad()
    # (no end-of-backward join is scheduled for a parked problem: whoever switched the queue on -- engine.StepBase -- ends the
    #  step's LAST backward() call with ``join_wgrad(force=True)``, which issues what is parked.  A join after every
    #  backward() call of a step made the next call's dX chain wait for weight gradients that feed only the optimizer:
    #  single-task step 0.915 -> 0.880 ms, multi-task steps neutral -- their heads' gradients were already parked this way)
    return True


def _take_parked(max_items: int):
    """Remove and return up to ``max_items`` parked weight-gradient problems (all of them or none: a partial take would
    reorder accumulations into one slot) for a caller that is about to issue a grouped launch of the same layout."""
    items = _wq["items"]
    if not items or len(items) > max_items:
        return []
    _wq["items"], _wq["tiles"] = [], 0
    return items


def _wgrad_defer_reduce(ws, dw, db, rows, cols, n_seg):
    """Park a norm layer's dw / db reduction (egk_ln_bwd_reduce arguments) with the next grouped launch: the parked
    reductions are issued as ONE launch (egk_ln_bwd_reduce_multi)."""
    _wq["extra"].append((ws, dw, db, rows, cols, n_seg))
    _wq["hold"].append(ws)


def _launch_reductions(reds):
    lib = _lib.load()
    # batches of at most 8 reductions with DISTINCT targets: a parameter used several times in one step (per-task backbone
    # passes) has several reductions accumulating into the same dw / db -- those stay in separate, ordered launches
    chunks, cur, seen = [], [], set()
    for r in reds:
        key = (r[1].data_ptr(), r[2].data_ptr())
        if len(cur) == 8 or key[0] in seen or key[1] in seen:
            chunks.append(cur)
            cur, seen = [], set()
        cur.append(r)
        seen.update(key)
    if cur:
        chunks.append(cur)
    for chunk in chunks:
        n = len(chunk)
        if n == 1:
            ws, dw, db, rows, cols, n_seg = chunk[0]
            _ck(lib.egk_ln_bwd_reduce(_stream(), _p(ws), _p(dw), _p(db), rows, cols, n_seg), "egk_ln_bwd_reduce")
            continue
        _ck(lib.egk_ln_bwd_reduce_multi(_stream(), _ptr_array([c[0] for c in chunk]), _ptr_array([c[1] for c in chunk]),
                                        _ptr_array([c[2] for c in chunk]), (C.c_int32 * n)(*[c[3] for c in chunk]),
                                        (C.c_int32 * n)(*[c[4] for c in chunk]), (C.c_int32 * n)(*[c[5] for c in chunk]), n),
            "egk_ln_bwd_reduce_multi")


# (experiment) a grouped weight-gradient launch holds every CU for the length of its K walk (6144 rows: 90-130 us) and a launch of
# the backward chain that arrives meanwhile waits for its workgroups to retire (a 43 us dX contraction took 135 us).
# EGK_WGRAD_KCHUNKS = n issues the group as n launches over consecutive K ranges, each accumulating into the gradient slots.
WGRAD_KCHUNKS = int(os.environ.get("EGK_WGRAD_KCHUNKS", "1"))


def _k_pieces(chunk):
    """``chunk`` (parked (args, kw) weight-gradient problems) cut into WGRAD_KCHUNKS lists over consecutive K ranges, or None."""
    n = WGRAD_KCHUNKS
    if n <= 1:
        return None
    K = chunk[0][0][6]
    for a, kw in chunk:
        if not (kw.get("transA") and kw.get("transB") and kw.get("accumulate") and a[6] == K and a[2].dim() == 2 and a[4].dim() == 2
                and a[2].shape[0] == K and a[4].shape[0] == K and kw.get("K2", 0) in (0, None)):
            return None
    step = (K // n + 63) // 64 * 64
    if step < 512:
        return None
    out = []
    for k0 in range(0, K, step):
        k1 = min(K, k0 + step)
        out.append([((a[0], a[1], a[2][k0:k1], a[3], a[4][k0:k1], a[5], k1 - k0, a[7], a[8]), kw) for a, kw in chunk])
    return out


def flush_wgrad(in_backward: bool = True, force: bool = False):
    """Issue what is parked.  On an excluded (task-head) stream nothing is issued -- unless ``force``: the engine's own
    calls from the backward stream (end of a step's backward, the last-weight-gradient hook) must never leave parked work
    behind, whatever stream handle the backward stream happens to have (a capture stream may alias a pooled handle that an
    earlier step registered as excluded)."""
    items, hold, extra, riders = _wq["items"], _wq["hold"], _wq["extra"], _wq.get("riders") or []
    if (not items and not extra and not riders) or (_on_excluded_stream() and not force):  # (a head stream never issues what others parked)
        return
    _wq["items"], _wq["hold"], _wq["extra"], _wq["tiles"], _wq["riders"] = [], [], [], 0, []
    rng = _last_wgrad["tail"]
    if rng is not None and (any(rng[0] <= it[0][7].data_ptr() < rng[1] for it in items)
                            or any(rng[0] <= e[1].data_ptr() < rng[1] or rng[0] <= e[2].data_ptr() < rng[1] for e in extra)):
        _last_wgrad["leaked"] = True  # (a gradient of the tail range leaves for the side stream: see last_wgrad_tail_on_backward_stream)

    def launch():
        stamp("wgrad_flush", seq=True)  # (on the side stream: when this flush's launches can start)
        for fn in riders:
            fn()
        for i in range(0, len(items), 8):
            chunk = items[i:i + 8]
            stamp("wgrad_group", seq=True)
            pieces = _k_pieces(chunk) if len(chunk) > 1 else None
            if pieces is not None:
                for piece in pieces:
                    gemm_grouped(piece, four_wave=len(piece) <= 4 and switches.enabled("wg4"))
            elif len(chunk) == 1:
                gemm(*chunk[0][0], **chunk[0][1])
            else:
                # beside the dX chain a group runs on 4-WAVE workgroups also when it has <= 256 tiles (alone the 8-wave
                # two-wave-group variant is faster there: 70 vs 85 us for four H x H problems): an 8-wave workgroup takes
                # 2 x 208 VGPRs of every SIMD of its CU and the chain's row kernels (96-193 VGPRs) wait for it to leave --
                # a 13 us row-LayerNorm backward took 82 us behind such a launch; 4 waves leave 304.  Step 1.511 -> 1.498 ms
                # (six alternating runs of 300 steps, every pair)
                gemm_grouped(chunk, four_wave=len(chunk) <= 4 and switches.enabled("wg4"))
        if extra:
            _launch_reductions(extra)
        stamp("wgrad_flush_end", seq=True)
    _wgrad_launch(True, hold, launch, in_backward)


_last_wgrad = {"param": None, "hook": None, "inline": True, "tail": None, "leaked": False}


def set_last_wgrad_tail(lo_ptr: int, hi_ptr: int) -> None:
    """With a last-weight-gradient hook installed: parked weight gradients and norm reductions whose gradient slots lie in
    the address range [lo_ptr, hi_ptr) are issued TOGETHER WITH the last weight gradient as one grouped launch on the
    backward stream (the caller's hook starts the optimizer on everything OUTSIDE that range meanwhile and steps the range
    after the launch).  Stand-alone, the first TRN linear's weight gradient alone takes 120 us and the two other TRN weight
    gradients as a group of their own 66 us; the three in one launch 123 us (tools/exp/tail_group_bench.py).
    (0, 0): only the last weight gradient itself (the default)."""
    _last_wgrad["tail"] = (int(lo_ptr), int(hi_ptr)) if hi_ptr > lo_ptr else None
    _last_wgrad["leaked"] = False


def last_wgrad_tail_on_backward_stream() -> bool:
    """Every gradient of the tail range (``set_last_wgrad_tail``) was issued on the BACKWARD stream: no parked problem or reduction
    of the range left with an earlier flush for the side stream, no weight gradient of the range was launched there directly.  The
    optimizer slice over the range may then follow the last weight gradient on the backward stream without waiting for the side
    stream's queue (engine.StepBase: the step's tail)."""
    return _last_wgrad["tail"] is not None and not _last_wgrad["leaked"]


def _take_tail_items():
    """Split what is parked by gradient-slot address: (contractions, reductions) inside the tail range are returned and
    removed from the queue; the rest stays parked."""
    rng = _last_wgrad["tail"]
    if rng is None:
        return [], []
    inside = lambda t: rng[0] <= t.data_ptr() < rng[1]
    items = [it for it in _wq["items"] if inside(it[0][7])]
    _wq["items"] = [it for it in _wq["items"] if not inside(it[0][7])]
    extra = [e for e in _wq["extra"] if inside(e[1]) and inside(e[2])]
    _wq["extra"] = [e for e in _wq["extra"] if not (inside(e[1]) and inside(e[2]))]
    _wq["tiles"] = sum(((it[0][0] + 127) // 128) * ((it[0][1] + 127) // 128) for it in _wq["items"])
    return items, extra


def set_last_wgrad_hook(param, hook, pre=None) -> None:
    """``hook()`` is called (on the backward stream) just before the weight-gradient launch of ``param`` -- the engine
    marks the LAST weight gradient of the step with it to start the optimizer on everything else meanwhile.  ``pre()`` runs
    first, before what is parked is issued: a step whose backward has branches on other streams that nothing has joined yet
    joins them there (parked problems of those branches read operands made on their streams)."""
    _last_wgrad["param"], _last_wgrad["hook"], _last_wgrad["pre"] = param, hook, pre
    if hook is None:
        _last_wgrad["tail"] = None


def wgrad_side_stream(main) -> Optional[torch.cuda.Stream]:
    """The weight-gradient side stream of ``main`` if one has been created."""
    return _wgrad["streams"].get((main.device.index, main.cuda_stream))


def _ln_reduce_on_side(slot_w, slot_b, x) -> bool:
    """The norm layers' dw / db reduction moves to the weight-gradient side stream when there is one for this stream
    (side streams on, stream not one of the excluded head / task streams) and both gradients accumulate in place."""
    if not (_wgrad["enabled"] and _wgrad_ln["side"] and slot_w is not None and slot_b is not None and x.is_cuda):
        return False
    main = torch.cuda.current_stream()
    return (main.device.index, main.cuda_stream) not in _wgrad["exclude"]


_wgrad_ln = {"side": True}  # development knob


def join_wgrad(force: bool = False):
    """The current stream waits for every weight-gradient side stream with work in flight (call after backward,
    before the gradients are read: optimizer step, gradient exchange, or the end of a hipGraph capture).  ``force``: the
    caller IS the step's backward stream (see flush_wgrad)."""
    if _wgrad.get("handoff") and not force:
        return  # (end-of-backward callback while the engine hands the side streams to its exchange: take_wgrad_streams)
    flush_wgrad(in_backward=False, force=force)
    drain_deferred()
    cur = torch.cuda.current_stream() if _wgrad["pending"] else None
    for side in _wgrad["pending"]:
        cur.wait_stream(side)
    _wgrad["pending"].clear()
    _wgrad["queued"] = False


def set_wgrad_handoff(on: bool) -> bool:
    """While on, the end of a backward() call does NOT make the backward stream wait for the weight-gradient side streams:
    the engine takes them over with ``take_wgrad_streams`` and orders its gradient exchange behind them instead, so the next
    backward stage's dX chain starts beside the weight gradients of the stage before.  Returns the previous setting."""
    prev = bool(_wgrad.get("handoff"))
    _wgrad["handoff"] = bool(on)
    return prev


def take_wgrad_streams() -> list:
    """Issue whatever is parked, then hand over the side streams with weight-gradient work in flight: the CALLER orders its
    consumer of the gradients behind them (and eventually joins that consumer into the backward stream)."""
    flush_wgrad(in_backward=False, force=True)
    drain_deferred()
    streams = list(_wgrad["pending"])
    _wgrad["pending"].clear()
    _wgrad["queued"] = False
    return streams


def _operand_rows(g: torch.Tensor, dtype: torch.dtype, pad_cols: int = 0) -> torch.Tensor:
    """[rows, cols] gradient as a contraction operand of element type ``dtype`` with a 16-byte aligned row stride.
    (autograd hands the gradient of an f32 output back as contiguous f32 whatever the loss emitted, and a width
    such as 478 or 115 is not a multiple of 8: one strided conversion instead of an unaligned, scalar-load GEMM)
    ``pad_cols`` > cols: the copy gets a row stride of pad_cols with ZEROS in [cols, pad_cols) -- the dX contraction
    of a classifier layer then runs over a K axis that is a multiple of 64 (``.base_cols`` of the result)."""
    esize = 2 if dtype == torch.bfloat16 else 4
    rows, cols = g.shape if g.dim() == 2 else (0, 0)
    want_pad = pad_cols > cols
    ok = g.dim() == 2 and g.stride(1) == 1 and (g.stride(0) * esize) % 16 == 0 and g.data_ptr() % 16 == 0
    if g.dtype == dtype and ok and not want_pad:
        return g
    if g.dim() != 2 or g.stride(1) != 1:
        g = g.contiguous()
    rows, cols = g.shape
    per16 = 16 // esize
    ld = pad_cols if want_pad else (cols + per16 - 1) // per16 * per16
    out = torch.empty((rows, ld), dtype=dtype, device=g.device)
    _ck(_lib.load().egk_cast_rows(_stream(), _p(g), _dt(g), g.stride(0), _p(out), _dt(out), ld, rows, cols,
                                  ld if want_pad else 0), "egk_cast_rows")
    return out[:, :cols]


def _match(g: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    g = _rm(g) if g.dim() == 2 else _c(g)
    return g if g.dtype == dtype else cast_raw(g, dtype)


# ---- Linear (two-source, fused bias / ReLU / residual) ----------------------------------------------
# ---- one forward pass for two consumers: the precise (three-product) values AND a bf16 autograd graph ---------------------------
# EgoPack's novel-task step needs the backbone twice: f32-grade features for the prototype search (an index op), and a bf16
# training graph for backward.  ``dual_record()`` tapes the results of the backbone's nodes while the PRECISE pass runs (no
# gradient); ``dual_replay(tape)`` then builds the bf16 graph by calling the same forward code, in which every node takes the
# rounding of the taped result (and the taped statistics) instead of launching its kernels: what backward saves are the bf16
# roundings of the precise activations, the forward VALUES are the precise ones.  Nodes: _Linear, _RowLN, _PEAdd, _SageMean, _GraphLN
# (what models.Graph.forward with a TRN pooling issues); the sequences of the two passes must agree node by node (checked).
_dual = {"tape": None, "replay": None}
_slab_defer = {"on": switches.enabled("slab_defer")}  # the precise pass's split contractions without reduce launches


class dual_record:
    """``with ops.dual_record() as tape:`` -- inside (the precise pass), the taped nodes append their results to ``tape``."""

    def __enter__(self):
        self.prev = _dual["tape"]
        _dual["tape"] = tape = []
        return tape

    def __exit__(self, *exc):
        _dual["tape"] = self.prev
        return False


class dual_replay:
    """``with ops.dual_replay(tape):`` -- inside (the bf16 pass, gradients on), the taped nodes take their results from ``tape``."""

    def __init__(self, tape):
        self.tape = list(tape)

    def __enter__(self):
        self.prev = _dual["replay"]
        _dual["replay"] = self.tape
        return self

    def __exit__(self, et, ev, tb):
        left = len(self.tape)
        _dual["replay"] = self.prev
        if et is None and left:
            raise RuntimeError(f"dual_replay: {left} taped node(s) were not consumed: the two passes issued different node sequences")
        return False


def _tape_put(kind: str, **tensors) -> None:
    """(precise pass) the results of one node; for every f32 matrix whose producer also stored its bf16 halves (the split tee, or a
    split some contraction asked for since) the high half rides along as '<name>:hi' -- the replay then needs no rounding launch --
    and an event of the producing stream, so that the replay of THIS node can run as soon as the node has."""
    if _dual["tape"] is None:
        return
    cache = _x3["cache"]
    if cache is not None:
        for k, t in list(tensors.items()):
            if t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous():
                hit = cache.get((t.data_ptr(), t.shape[0], t.shape[1], t.shape[1], t._version))
                if hit is not None:
                    tensors[k + ":hi"] = hit[0]
    for t in list(tensors.values()):
        if getattr(t, "_egk_slabs", None) is not None:
            t._egk_tape_entry = tensors  # (its reduced values are written later: ``_slabs_written`` moves the event there)
    tensors["@event"] = torch.cuda.current_stream().record_event() if next(iter(tensors.values())).is_cuda else None
    _dual["tape"].append((kind, tensors))


def _tape_take(kind: str, like: torch.Tensor):
    """The taped results of the next node (None outside a replay); ``like``: the node's bf16 input -- shape / type guard."""
    rp = _dual["replay"]
    if rp is None:
        return None
    if not rp:
        raise RuntimeError(f"dual_replay: a '{kind}' node has no taped counterpart")
    k, t = rp.pop(0)
    if k != kind or like.dtype != torch.bfloat16:
        raise RuntimeError(f"dual_replay: node '{kind}' ({like.dtype}) met the taped '{k}': the two passes disagree")
    ev = t.get("@event")
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
    return t


def _r16(taped, name: str) -> torch.Tensor:
    """bf16 rounding (nearest-even: the 'hi' half of the three-product split) of the taped f32 result ``name``: the half its
    producer stored, or one rounding launch."""
    hi = taped.get(name + ":hi")
    return hi if hi is not None else cast_raw(_c(taped[name]), torch.bfloat16)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, x2, W2, residual, relu, compute, out_f32, ln_in=None, res_sink=None, slab_ok=False):
        _need_gpu(x, W)
        if not (compute == X3 and getattr(x, "_egk_virtual", False) and x.is_contiguous()):
            x = _c(x)  # (a lazily widened bf16 input stays unwritten: the three-product contraction reads its bf16 source)
        M, K1 = x.shape
        N = W.shape[0]
        Wop = weight_operand(W, x.dtype)
        K2 = 0
        W2op = None
        if x2 is not None:
            x2 = _match(x2, x.dtype)
            K2 = x2.shape[1]
            W2op = weight_operand(W2, x.dtype)
        res = _c(residual) if residual is not None else None
        taped = _tape_take("linear", x) if not out_f32 else None
        if taped is not None:
            y = _r16(taped, "y")
            if tuple(y.shape) != (M, N):
                raise RuntimeError("dual_replay: a taped linear result has another shape")
        else:
            if out_f32 and N % 8:  # logits: pad the row stride so the loss gradient is a 16-byte aligned operand
                y = torch.empty((M, _pad8(N)), dtype=torch.float32, device=x.device)[:, :N]
            else:
                y = torch.empty((M, N), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
            bias_c = _f32c(b) if b is not None else None
            # the precise pass (no gradient, inside a precise_scope): a launch that splits K leaves its slabs for the row kernel
            # that reads the result next (row LayerNorm, graph LayerNorm, PE add: ``_slab_consumer``) instead of a reduce launch
            # -- ONLY where the caller says that such a kernel is what reads the result (``slab_ok``): an unreduced tensor must never
            # reach anything that reads memory without asking this module for the pointer (torch operators, .cpu())
            defer = (slab_ok and compute == X3 and _slab_defer["on"] and _x3["cache"] is not None and not torch.is_grad_enabled()
                     and not relu and res is None and y.dtype == torch.float32 and y.is_contiguous() and N <= 1024 and N % 4 == 0)
            ws = gemm(M, N, x, K1, Wop, K1, K1, y, y.stride(0), A2=x2, lda2=K2, B2=W2op, ldb2=K2, K2=K2, bias=bias_c, residual=res,
                      ldr=N, act=1 if relu else 0, compute=compute, defer_reduce=defer)
            if ws is not None:
                _slab_mark(y, ws, M, N, bias_c)
            if not out_f32:
                _tape_put("linear", y=y)
        ctx.relu, ctx.compute = relu, compute
        ctx.ln_in = ln_in if (x2 is None and not relu) else None
        ctx.res_sink = res_sink if residual is not None else None
        ctx.has = (b is not None, x2 is not None, residual is not None)
        ctx.res_dtype = residual.dtype if residual is not None else None
        ctx.params = (W, b, W2)
        # classifier layers (N = 478, 115, 2 ...): the optimizer's flat layout keeps the bf16 copy of W with its rows
        # zero-padded to a multiple of 64, so dX = dY @ W can contract over a padded K on the pipelined kernel
        wpad = getattr(W, "_egk_shadow_rows64", None)
        ctx.Wpad = wpad if (wpad is not None and wpad.dtype == x.dtype and x2 is None and not relu) else None
        ctx.save_for_backward(x, Wop, x2, W2op, y if relu else None)
        if out_f32:
            y._egk_grad_dtype = x.dtype  # lets the loss emit its gradient in the operand type
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, x2, W2, y = ctx.saved_tensors
        Wp, bp, W2p = ctx.params
        has_b, has_x2, has_res = ctx.has
        Wpad = ctx.Wpad if ctx.needs_input_grad[0] else None
        g = _operand_rows(dy, x.dtype, pad_cols=Wpad.shape[0] if Wpad is not None else 0)
        M, N = g.shape
        K1 = x.shape[1]
        lib = _lib.load()
        if ctx.relu:
            assert not has_res
            g = g.contiguous()
            gg = torch.empty_like(g)
            _ck(lib.egk_relu_gate(_stream(), _p(g), _p(_match(y, g.dtype)), _p(gg), g.numel(), _dt(g)), "egk_relu_gate")
            g = gg
        needs = ctx.needs_input_grad
        dx = dW = db = dx2 = dW2 = None
        if needs[0]:
            dx = torch.empty_like(x)
            if Wpad is not None and g.stride(0) == Wpad.shape[0]:  # zero columns of g x zero rows of the padded copy
                gemm(M, K1, g, g.stride(0), Wpad, K1, Wpad.shape[0], dx, K1, transB=True, compute=ctx.compute)
            elif ctx.ln_in is not None:  # dx is dy of the graph LayerNorm that produced x: its segment sums ride on this launch
                _ln_bwd_stats_launch(ctx.ln_in, (M, K1, g, g.stride(0), W, K1, N, dx, K1), dict(transB=True, compute=ctx.compute), dx)
            else:
                gemm(M, K1, g, g.stride(0), W, K1, N, dx, K1, transB=True, compute=ctx.compute)
        db_out = None
        if has_b and needs[2]:
            slot_b = _grad_slot(bp)
            db_out = slot_b if slot_b is not None else torch.zeros(N, dtype=torch.float32, device=g.device)
            db = None if slot_b is not None else db_out
        if has_x2 and needs[3]:
            K2 = x2.shape[1]
            dx2 = torch.empty_like(x2)
            gemm(M, K2, g, g.stride(0), W2, K2, N, dx2, K2, transB=True, compute=ctx.compute)
        if needs[1]:
            slot = _grad_slot(Wp)
            out = slot if slot is not None else torch.zeros(W.shape, dtype=torch.float32, device=g.device)
            last = Wp is _last_wgrad["param"] and _last_wgrad["hook"] is not None
            tail_items, tail_extra = [], []
            if last:
                if _last_wgrad.get("pre") is not None:
                    _last_wgrad["pre"]()
                tail_items, tail_extra = _take_tail_items()  # (issued below; the argument tuples keep their operands alive)
                flush_wgrad(force=True)  # (the hook starts the optimizer on every other slot: their gradients must be issued)
                _last_wgrad["hook"]()
            # the bias gradient colsum(dY) rides on the dW launch (summed from the dY^T tile already in LDS).  The LAST
            # weight gradient of the step stays on the backward stream: nothing is left to overlap it with there, and the
            # hop to the side stream and back costs two cross-queue hand-offs (~10 us each) on the step's tail.
            in_place = slot is not None and (db_out is None or db is None)
            dw_args = (N, K1, g, g.stride(0), x, K1, M, out, K1)
            dw_kw = dict(transA=True, transB=True, accumulate=True, compute=ctx.compute, dbias=db_out)
            if last and (tail_items or tail_extra) and in_place and ctx.compute in (BF16, F32):
                # the step's tail as ONE grouped launch on the backward stream: the parked weight gradients of the tail range
                # + this one (the largest last), then the norm reductions of the range
                if tail_items:
                    gemm_grouped(tail_items + [(dw_args, dw_kw)])
                else:
                    gemm(*dw_args, **dw_kw)
                if tail_extra:
                    _launch_reductions(tail_extra)
            elif not (in_place and not last and ctx.compute in (BF16, F32) and _wgrad_defer(dw_args, dw_kw, (g, x))):
                rng = _last_wgrad["tail"]
                if rng is not None and not last and rng[0] <= out.data_ptr() < rng[1]:
                    _last_wgrad["leaked"] = True  # (launched below on the side stream)
                for it in tail_items:  # (not eligible after all: issue what was taken, in order)
                    gemm(*it[0], **it[1])
                if tail_extra:
                    _launch_reductions(tail_extra)
                _wgrad_launch(in_place and not (last and _last_wgrad["inline"]), (g, x), lambda: gemm(*dw_args, **dw_kw))
            dW = None if slot is not None else out
        elif db_out is not None:
            _colsum_into(g, db_out, True)
        if has_x2 and needs[4]:
            K2 = x2.shape[1]
            slot = _grad_slot(W2p)
            out2 = slot if slot is not None else torch.zeros(W2.shape, dtype=torch.float32, device=g.device)
            _wgrad_launch(slot is not None, (g, x2),
                          lambda: gemm(N, K2, g, g.stride(0), x2, K2, M, out2, K2, transA=True, transB=True,
                                       accumulate=True, compute=ctx.compute))
            dW2 = None if slot is not None else out2
        dres = _match(dy, ctx.res_dtype) if (has_res and needs[5]) else None
        if dres is not None and ctx.res_sink is not None:
            # the residual operand's gradient is handed to the node named by the caller (models.Graph: the first SAGE layer,
            # which adds it in its dX epilogue) instead of to autograd, which would add the two gradients of x in a pass
            ctx.res_sink["dy"] = dres
            dres = None
        return dx, dW, db, dx2, dW2, dres, None, None, None, None, None, None


def _compute_for(x):
    return BF16 if x.dtype == torch.bfloat16 else _state["compute"]


# ---- classifier bank: several Linear layers over the same input as one contraction ---------------------------------------
_bank_handoff = {"on": False}


class bank_grad_handoff:
    """``with bank_grad_handoff():`` -- inside, a cross-entropy whose logits come straight from ``classifier_bank`` writes
    its gradient into the bank's padded operand buffer (element type and row stride the dX / dW contractions want) and
    hands autograd an uninitialised placeholder, which the bank's backward ignores.  Legal only where every such logits
    tensor feeds exactly ONE loss node and nothing else that needs its gradient (the engine's training heads)."""

    def __enter__(self):
        self.prev, _bank_handoff["on"] = _bank_handoff["on"], True

    def __exit__(self, *a):
        _bank_handoff["on"] = self.prev


def _bank_gradient_operand(views, gbuf, state, gs, x):
    """The [M, sum rows64] gradient operand of a classifier bank's backward contractions: the columns the loss wrote itself
    (bank_grad_handoff) stay, the others are converted from the incoming gradients ``gs``, pad columns are zero."""
    lib = _lib.load()
    M, N = x.shape[0], views["n"]
    if gbuf is None:  # (forward ran without a gradient consumer in sight; cannot happen under autograd)
        gbuf = torch.zeros((M, N), dtype=x.dtype, device=x.device)
    if not state["pads"]:  # allocated uncleared for the fused loss, which then did not run: clear the pads now
        ends = [r for r, _ in views["rows"][1:]] + [N]
        for (r0, n), e in zip(views["rows"], ends):
            if e > r0 + n:
                gbuf[:, r0 + n:e].zero_()
            if r0 not in state["filled"]:
                gbuf[:, r0:r0 + n].zero_()
    for (r0, n), g in zip(views["rows"], gs):
        if r0 in state["filled"] or g is None:
            continue  # the loss wrote these columns itself (bank_grad_handoff); None: no gradient, columns stay zero
        g = _rm(g)
        dst = gbuf[:, r0:r0 + n]
        _ck(lib.egk_cast_rows(_stream(), _p(g), _dt(g), g.stride(0), _p(dst), _dt(gbuf), gbuf.stride(0), M, n, 0),
            "egk_cast_rows")
    return gbuf


class _ClassifierBank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, views, gbuf, state, compute):
        _need_gpu(x)
        x = _c(x)
        M, K = x.shape
        Wop = views["w16"] if x.dtype == torch.bfloat16 else views["wp"]
        out = torch.empty((M, views["n"]), dtype=torch.float32, device=x.device)
        gemm(M, views["n"], x, K, Wop, K, K, out, views["n"], bias=views["b"], compute=compute)
        ctx.views, ctx.gbuf, ctx.state, ctx.compute = views, gbuf, state, compute
        ctx.save_for_backward(x, Wop)
        return tuple(out[:, r0:r0 + n] for r0, n in views["rows"])

    @staticmethod
    def backward(ctx, *gs):
        x, Wop = ctx.saved_tensors
        views, gbuf, lib = ctx.views, ctx.gbuf, _lib.load()
        M, K = x.shape
        N = views["n"]
        gbuf = _bank_gradient_operand(views, gbuf, ctx.state, gs, x)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(M, K, gbuf, N, Wop, K, N, dx, K, transB=True, compute=ctx.compute)
        w_args = (N, K, gbuf, N, x, K, M, views["wg"], K)
        w_kw = dict(transA=True, transB=True, accumulate=True, compute=ctx.compute, dbias=views["bg"])
        if not (ctx.compute == BF16 and _wgrad_defer(w_args, w_kw, (gbuf, x), park_on_excluded=True)):
            _wgrad_launch(True, (gbuf, x), lambda: gemm(*w_args, **w_kw))
        return dx, None, None, None, None, None


def classifier_bank(x, anchor, views, compute=None):
    """Logits of every classifier of a bank (``views``: optim.FlatAdam._bank_views) as column ranges of ONE
    [rows, sum rows64] f32 contraction output; backward is one dX and one dW contraction over the zero-padded bank."""
    needs = torch.is_grad_enabled() and (x.requires_grad or anchor.requires_grad)
    # the gradient operand: zero pad columns.  Under a known loss seed (loss_seed) the fused cross entropy writes every
    # column of every block itself, pads included, so the buffer is not cleared first (state["pads"] records that it did)
    lazy = _bank_handoff["on"] and _loss_seed["coef"] is not None
    gbuf = None
    if needs:
        gbuf = (torch.empty if lazy else torch.zeros)((x.shape[0], views["n"]), dtype=x.dtype, device=x.device)
    state = {"filled": set(), "pads": not lazy}
    outs = _ClassifierBank.apply(x, anchor, views, gbuf, state, _compute_for(x) if compute is None else compute)
    if _bank_handoff["on"] and gbuf is not None:
        for o, (r0, _) in zip(outs, views["rows"]):
            o._egk_grad_dst = (gbuf, r0, state)
    return outs


_banks_ride = {"on": switches.enabled("banks_ride")}  # development knob


class _GroupedBanks(torch.autograd.Function):
    """The classifier banks of several task heads (same feature width, own rows, own weights) as ONE grouped contraction
    forward and ONE grouped dX contraction backward -- the multi-head classifiers of the AR and LTA tasks
    (models/tasks/recognition.py:37-45, called per task at main_temporal.py:93-126) -- instead of one chain per task on
    its own stream.  The weight gradients are parked like those of ``_ClassifierBank``."""

    @staticmethod
    def forward(ctx, views_list, gbufs, states, compute, *xs):
        xs = [_c(x) for x in xs]
        outs, probs = [], []
        for x, views in zip(xs, views_list):
            M, K = x.shape
            out = torch.empty((M, views["n"]), dtype=torch.float32, device=x.device)
            probs.append(((M, views["n"], x, K, views["w16"], K, K, out, views["n"]), dict(bias=views["b"], compute=compute)))
            outs.append(out)
        gemm_grouped(probs)
        ctx.views_list, ctx.gbufs, ctx.states, ctx.compute = views_list, gbufs, states, compute
        ctx.save_for_backward(*xs)
        return tuple(out[:, r0:r0 + n] for out, views in zip(outs, views_list) for r0, n in views["rows"])

    @staticmethod
    def backward(ctx, *gs):
        xs = ctx.saved_tensors
        cmp, k, probs, dxs, parked = ctx.compute, 0, [], [], []
        for i, (x, views) in enumerate(zip(xs, ctx.views_list)):
            nb = len(views["rows"])
            gbuf = _bank_gradient_operand(views, ctx.gbufs[i], ctx.states[i], gs[k:k + nb], x)
            k += nb
            M, K = x.shape
            N = views["n"]
            if ctx.needs_input_grad[4 + i]:
                dx = torch.empty_like(x)
                probs.append(((M, K, gbuf, N, views["w16"], K, N, dx, K), dict(transB=True, compute=cmp)))
                dxs.append(dx)
            else:
                dxs.append(None)
            parked.append(((N, K, gbuf, N, x, K, M, views["wg"], K),
                           dict(transA=True, transB=True, accumulate=True, compute=cmp, dbias=views["bg"]), (gbuf, x)))
        if len(probs) > 1:
            gemm_grouped(probs)
        elif probs:
            gemm(*probs[0][0], **probs[0][1])
        for w_args, w_kw, hold in parked:
            # parked WITHOUT a join of their own: the projection heads' backward, which follows in the same step, takes them
            # along in its first grouped weight-gradient launch (one launch and one end-of-backward join fewer in the chain)
            if not _wgrad_defer(w_args, w_kw, hold, park_on_excluded=True, park_only=_banks_ride["on"]):
                _wgrad_launch(True, hold, lambda a=w_args, kw=w_kw: gemm(*a, **kw))
        return (None, None, None, None, *dxs)


def grouped_classifier_banks_ok(xs, views_list) -> bool:
    """bf16 activations, 2 .. 8 banks over features of one width (a multiple of 64), row counts multiples of 64."""
    if not (2 <= len(xs) <= 8) or any(v is None for v in views_list):
        return False
    K = xs[0].shape[1] if xs[0].dim() == 2 else 0
    for x, v in zip(xs, views_list):
        if (x.dim() != 2 or not x.is_cuda or x.dtype != torch.bfloat16 or x.shape[1] != K or K % 64 or x.shape[0] % 64
                or x.shape[0] == 0 or v["n"] % 64 or v.get("w16") is None):
            return False
    return True


def grouped_classifier_banks(xs, views_list, fused_loss: bool = False, compute=None):
    """Per bank, the tuple of logits ``classifier_bank`` returns -- all banks in one launch (see ``_GroupedBanks``).
    ``fused_loss``: a cross entropy with an announced seed follows for every bank (it then writes the whole gradient
    operand itself, so the operand is not cleared first)."""
    needs = torch.is_grad_enabled()
    lazy = _bank_handoff["on"] and fused_loss
    gbufs, states = [], []
    for x, views in zip(xs, views_list):
        gbufs.append((torch.empty if lazy else torch.zeros)((x.shape[0], views["n"]), dtype=x.dtype, device=x.device) if needs else None)
        states.append({"filled": set(), "pads": not lazy})
    flat = _GroupedBanks.apply(list(views_list), gbufs, states, _compute_for(xs[0]) if compute is None else compute, *xs)
    outs, k = [], 0
    for views, gbuf, state in zip(views_list, gbufs, states):
        blk = flat[k:k + len(views["rows"])]
        k += len(views["rows"])
        if _bank_handoff["on"] and gbuf is not None:
            for o, (r0, _) in zip(blk, views["rows"]):
                o._egk_grad_dst = (gbuf, r0, state)
        outs.append(tuple(blk))
    return outs


def linear(x, W, b=None, *, x2=None, W2=None, residual=None, relu=False, compute=None, out_f32=False, ln_in=None, res_sink=None,
           slab_ok=False):
    """y = relu?(x @ W.T (+ x2 @ W2.T) + b) (+ residual): one MFMA launch.  ``out_f32`` keeps the result
    in f32 whatever the activation type (logits).  ``ln_in``: the context of the graph LayerNorm that produced x
    (``graph_layernorm_lrelu(..., return_ctx=True)``): its backward sums are then taken in this layer's dX epilogue.
    ``res_sink``: a dict -- in backward the gradient of ``residual`` is put there (key 'dy') instead of being returned to
    autograd; the caller guarantees that a node earlier in the graph (``sage_mean_layer(..., res_src=)``) adds it.
    ``slab_ok``: the caller hands the result STRAIGHT to ``row_layernorm`` / ``graph_layernorm_lrelu`` / ``pe_add`` and to nothing
    else -- in the forward-only precise pass a launch that splits K may then leave its slabs to that kernel (``_slab_consumer``)."""
    return _Linear.apply(x, W, b, x2, W2, residual, relu, _compute_for(x) if compute is None else compute, out_f32, ln_in, res_sink,
                         bool(slab_ok))


class _MultiLinear(torch.autograd.Function):
    """y = cat_i(x_i) @ W.T + b without materialising the concatenation: one contraction per input
    block writing its row slice of the shared output (the 1536-d feature rows of every task batch are
    read where the loader put them)."""

    @staticmethod
    def forward(ctx, W, b, compute, *xs):
        _need_gpu(W, *xs)
        xs = [_c(x) for x in xs]
        dt = xs[0].dtype
        Wop = weight_operand(W, dt)
        N, K = Wop.shape
        rows = [x.shape[0] for x in xs]
        y = torch.empty((sum(rows), N), dtype=dt, device=W.device)
        bc = _f32c(b) if b is not None else None
        off = 0
        for x, m in zip(xs, rows):
            gemm(m, N, x, K, Wop, K, K, y[off:off + m], N, bias=bc, compute=compute)
            off += m
        ctx.compute, ctx.rows, ctx.params = compute, rows, (W, b)
        ctx.save_for_backward(Wop, *xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        Wop, *xs = ctx.saved_tensors
        Wp, bp = ctx.params
        dy = _match(dy, xs[0].dtype)
        N, K = Wop.shape
        needs = ctx.needs_input_grad
        dW = db = None
        dxs = []
        off = 0
        for i, (x, m) in enumerate(zip(xs, ctx.rows)):
            if needs[3 + i]:
                dx = torch.empty_like(x)
                gemm(m, K, dy[off:off + m], N, Wop, K, N, dx, K, transB=True, compute=ctx.compute)
                dxs.append(dx)
            else:
                dxs.append(None)
            off += m
        if needs[0]:
            slot = _grad_slot(Wp)
            out = slot if slot is not None else torch.zeros(Wop.shape, dtype=torch.float32, device=dy.device)

            def launch_dw():
                off = 0
                for x, m in zip(xs, ctx.rows):
                    gemm(N, K, dy[off:off + m], N, x, K, m, out, K, transA=True, transB=True, accumulate=True,
                         compute=ctx.compute, slot_final=len(xs) == 1)  # (several launches add up one gradient: not final in any of them)
                    off += m
            _wgrad_launch(slot is not None, (dy, *xs), launch_dw)
            dW = None if slot is not None else out
        if bp is not None and needs[1]:
            slot = _grad_slot(bp)
            outb = slot if slot is not None else torch.zeros(N, dtype=torch.float32, device=dy.device)
            _wgrad_launch(slot is not None, (dy,), lambda: _colsum_into(dy, outb, True))
            db = None if slot is not None else outb
        return (dW, db, None, *dxs)


def multi_linear(xs, W, b=None, compute=None):
    return _MultiLinear.apply(W, b, _compute_for(xs[0]) if compute is None else compute, *xs)


# ---- grouped projection heads: Linear -> LayerNorm -> ReLU -> Linear of several tasks as ONE chain of grouped launches ----
def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _GroupedProjection(torch.autograd.Function):
    """``ProjectionTask.net`` (reference models/tasks/task.py:17-26 with dropout 0) of G task batches: every stage is one
    launch over all groups -- grouped contraction, grouped row LayerNorm + ReLU, grouped contraction forward; the two
    grouped dX contractions, the grouped LayerNorm backward and the two grouped dW (+ fused bias gradient) launches
    backward -- instead of G chains on G streams.  Inputs: G row blocks x_g [M_g, H]; per group (W1, b1, lnw, lnb, W2, b2).
    Parameter gradients accumulate in place (the optimizer's flat gradient slots must exist)."""

    @staticmethod
    def forward(ctx, G, compute, eps, specs, *args):
        lib = _lib.load()
        full = [_c(x) for x in args[:G]]
        # a group with a row spec (first, step, count, inv) consumes rows first + i * step, i < count, of its input -- a strided view,
        # no copy: the contractions take any row stride -- and its feature gradient goes back to full height through ``inv``
        xs = [x if sp is None else x[sp[0]::sp[1]][:sp[2]] for x, sp in zip(full, specs)]
        params = [args[G + 6 * g: G + 6 * g + 6] for g in range(G)]
        dt, dev = xs[0].dtype, xs[0].device
        rows = [x.shape[0] for x in xs]
        H = xs[0].shape[1]
        H1, H2 = params[0][0].shape[0], params[0][4].shape[0]
        ptr = [0]
        for m in rows:
            ptr.append(ptr[-1] + m)
        n = ptr[-1]
        W1o = [weight_operand(p[0], dt) for p in params]
        W2o = [weight_operand(p[4], dt) for p in params]
        h1 = torch.empty((n, H1), dtype=dt, device=dev)
        a = torch.empty_like(h1)
        f = torch.empty((n, H2), dtype=dt, device=dev)
        mean = torch.empty(n, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        gemm_grouped([((rows[g], H1, xs[g], xs[g].stride(0), W1o[g], H, H, h1[ptr[g]:ptr[g + 1]], H1),
                       dict(bias=_f32c(params[g][1]), compute=compute)) for g in range(G)])
        lw, lb = [_f32c(p[2]) for p in params], [_f32c(p[3]) for p in params]
        row_ptr = (C.c_int32 * (G + 1))(*ptr)
        _ck(lib.egk_rowln_group_fwd(_stream(), _p(h1), _ptr_array(lw), _ptr_array(lb), row_ptr, G, _p(a), _p(mean), _p(rstd), H1,
                                    eps, 1, _dt(h1)), "egk_rowln_group_fwd")
        gemm_grouped([((rows[g], H2, a[ptr[g]:ptr[g + 1]], H1, W2o[g], H1, H1, f[ptr[g]:ptr[g + 1]], H2),
                       dict(bias=_f32c(params[g][5]), compute=compute)) for g in range(G)])
        ctx.G, ctx.compute, ctx.ptr, ctx.dims, ctx.params = G, compute, ptr, (H, H1, H2), params
        ctx.specs, ctx.full_rows = specs, [x.shape[0] for x in full]
        ctx.save_for_backward(h1, a, mean, rstd, *full, *W1o, *W2o, *lw, *lb, *[sp[3] for sp in specs if sp is not None])
        return tuple(f[ptr[g]:ptr[g + 1]] for g in range(G))

    @staticmethod
    def backward(ctx, *dfs):
        lib = _lib.load()
        G, ptr, (H, H1, H2), params = ctx.G, ctx.ptr, ctx.dims, ctx.params
        h1, a, mean, rstd, *rest = ctx.saved_tensors
        full, W1o, W2o, lw, lb = (rest[i * G:(i + 1) * G] for i in range(5))
        specs, invs = ctx.specs, list(rest[5 * G:])
        xs = [x if sp is None else x[sp[0]::sp[1]][:sp[2]] for x, sp in zip(full, specs)]
        dt, dev = h1.dtype, h1.device
        rows = [ptr[g + 1] - ptr[g] for g in range(G)]
        dfs = [_operand_rows(d if d is not None else torch.zeros((rows[g], H2), dtype=dt, device=dev), dt) for g, d in enumerate(dfs)]
        slots = [[_grad_slot(p) for p in pg] for pg in params]
        if any(s_ is None for sg in slots for s_ in sg):
            raise RuntimeError("grouped_projection: the parameters need in-place gradient slots (optim.FlatAdam materialised)")
        cmp = ctx.compute
        da = torch.empty_like(a)
        gemm_grouped([((rows[g], H1, dfs[g], dfs[g].stride(0), W2o[g], H1, H2, da[ptr[g]:ptr[g + 1]], H1),
                       dict(transB=True, compute=cmp)) for g in range(G)])
        proj_park = switches.enabled("proj_park")
        dw2 = [((H2, H1, dfs[g], dfs[g].stride(0), a[ptr[g]:ptr[g + 1]], H1, rows[g], slots[g][4], H1),
                dict(transA=True, transB=True, accumulate=True, compute=cmp, dbias=slots[g][5])) for g in range(G)]
        proj_park = proj_park and all(_wgrad_groupable(*pa[:7]) for pa, _ in dw2)  # (the parking queue must be on and take them)
        if proj_park:
            for g, (pa, pk) in enumerate(dw2):
                if not _wgrad_defer(pa, pk, (dfs[g], a), park_on_excluded=True, park_only=True):
                    raise RuntimeError("grouped_projection: a weight gradient announced as parkable was refused")
        else:
            riders = _take_parked(8 - G) if _banks_ride["on"] else []  # (the classifier banks' weight gradients, parked by their backward)
            _wgrad_launch(True, (a, *dfs), lambda: gemm_grouped(riders + dw2))
        dh1 = torch.empty_like(h1)
        grid = lib.egk_rowln_bwd_ws_rows(max(rows))
        ws = torch.empty(G * grid * 2 * H1 * 4, dtype=torch.uint8, device=dev)  # (own buffer: reduced on the side stream)
        row_ptr = (C.c_int32 * (G + 1))(*ptr)
        _ck(lib.egk_rowln_group_bwd(_stream(), _p(da), _p(h1), _ptr_array(lw), _ptr_array(lb), row_ptr, G, _p(mean), _p(rstd),
                                    _p(dh1), _p(ws), H1, 1, _dt(h1)), "egk_rowln_group_bwd")

        reds = [(ws[g * grid * 2 * H1 * 4:], slots[g][2], slots[g][3], max(rows), H1, 0) for g in range(G)]
        if proj_park:
            for r in reds:
                _wgrad_defer_reduce(*r)
        else:
            _wgrad_launch(True, (ws,), lambda: _launch_reductions(reds))
        dxs = [None] * G
        if any(ctx.needs_input_grad[4:4 + G]):
            # ONE buffer for the feature gradients of all groups at their FULL heights, consecutive (the fused backbone pass takes
            # it back without a copy, _SplitRows.backward); a group with a row spec gets its dx rows in a scratch block and one
            # launch puts them at their places with zero rows elsewhere (egk_gather_rows through the inverse map)
            fptr = [0]
            for n_ in ctx.full_rows:
                fptr.append(fptr[-1] + n_)
            dx = torch.empty((fptr[-1], H), dtype=dt, device=dev)
            scratch = {g: torch.empty((rows[g], H), dtype=dt, device=dev) for g in range(G) if specs[g] is not None}
            dst = [scratch[g] if g in scratch else dx[fptr[g]:fptr[g + 1]] for g in range(G)]
            gemm_grouped([((rows[g], H, dh1[ptr[g]:ptr[g + 1]], H1, W1o[g], H, H1, dst[g], H),
                           dict(transB=True, compute=cmp)) for g in range(G)])
            k_inv = 0
            for g in range(G):
                if g in scratch:
                    inv = invs[k_inv]
                    k_inv += 1
                    blk = dx[fptr[g]:fptr[g + 1]]
                    _ck(lib.egk_gather_rows(_stream(), _p(scratch[g]), _dt(dx), H, rows[g], _p(inv), _p(blk), _dt(dx), blk.shape[0], H),
                        "egk_gather_rows")
            dxs = [dx[fptr[g]:fptr[g + 1]] for g in range(G)]
        dw1 = [((H1, H, dh1[ptr[g]:ptr[g + 1]], H1, xs[g], xs[g].stride(0), rows[g], slots[g][0], H),
                dict(transA=True, transB=True, accumulate=True, compute=cmp, dbias=slots[g][1])) for g in range(G)]
        if proj_park and all(_wgrad_groupable(*pa[:7]) for pa, _ in dw1):
            for g, (pa, pk) in enumerate(dw1):
                if not _wgrad_defer(pa, pk, (dh1, xs[g]), park_on_excluded=True, park_only=True):
                    raise RuntimeError("grouped_projection: a weight gradient announced as parkable was refused")
        else:
            _wgrad_launch(True, (dh1, *xs), lambda: gemm_grouped(dw1))
        return (None, None, None, None, *dxs, *([None] * (6 * G)))


@torch.no_grad()
def grouped_projection_infer(x, nets, out_f32: bool = False):
    """f_g = net_g(x) for several projection heads over the SAME input rows, without autograd: three grouped launches
    (contraction, row LayerNorm + ReLU, contraction) instead of three per head -- the DETACHED auxiliary-task projections of
    the EgoPack step (reference main_egopack.py:121-147: ``tasks[t].forward_features(feat)`` under no_grad for every
    auxiliary task).  ``out_f32``: the last contraction keeps its f32 accumulators (the prototype search ranks those).
    None when the heads do not qualify (bf16 activations, equal widths in multiples of 64, 2 .. 8 heads, no active dropout)."""
    G = len(nets)
    # f32 activations inside a precise_scope (compute mode 'bf16x3': every contraction as three bf16 products of split operands):
    # the same three grouped launches with f32 in between, + ONE split launch for the second contraction's operand block
    x3 = x.dtype == torch.float32 and _state["compute"] == X3 and _x3["cache"] is not None and out_f32
    if not (2 <= G <= 8) or x.dim() != 2 or not x.is_cuda or not (x.dtype == torch.bfloat16 or x3) or x.shape[0] == 0:
        return None
    if x3 and not switches.enabled("x3_grouped_aux"):
        return None
    dims = None
    for net in nets:
        if len(net) != 5 or (net[0].p > 0 and net[0].training):
            return None
        l1, l2 = net[1], net[4]
        d = (l1.in_features, l1.out_features, l2.out_features)
        dims = dims or d
        if d != dims or l2.in_features != d[1] or x.shape[1] != d[0] or any(v % 64 for v in d) or d[1] > 4096:
            return None
        if l1.bias is None or l2.bias is None:
            return None
    lib = _lib.load()
    x = _c(x)
    M, (H, H1, H2) = x.shape[0], dims
    if x3:
        if M % 8 or H % 64 or H1 % 64:
            return None
        h1 = torch.empty((G * M, H1), dtype=torch.float32, device=x.device)
        a = torch.empty_like(h1)
        f = torch.empty((G * M, H2), dtype=torch.float32, device=x.device)
        mean = torch.empty(G * M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        gemm_grouped([((M, H1, x, H, net[1].weight, H, H, h1[g * M:(g + 1) * M], H1), dict(bias=_f32c(net[1].bias), compute=X3))
                      for g, net in enumerate(nets)])
        lw, lb = [_f32c(net[2].weight) for net in nets], [_f32c(net[2].bias) for net in nets]
        row_ptr = (C.c_int32 * (G + 1))(*[g * M for g in range(G + 1)])
        # the LayerNorm launch also stores the halves of its result (egk_tee_split_next; EGK_DISABLE=group_ln_tee: a split launch)
        halves = _tee_arm(a) if switches.enabled("group_ln_tee") else None
        _ck(lib.egk_rowln_group_fwd(_stream(), _p(h1), _ptr_array(lw), _ptr_array(lb), row_ptr, G, _p(a), _p(mean), _p(rstd), H1,
                                    float(nets[0][2].eps), 1, _dt(h1)), "egk_rowln_group_fwd")
        hi, lo = halves if halves is not None else _split_rows(a, G * M, H1, H1)  # the groups' row blocks are registered as split
        for g in range(G):
            ag = a[g * M:(g + 1) * M]
            _x3["cache"][(ag.data_ptr(), M, H1, H1, ag._version)] = (hi[g * M:(g + 1) * M], lo[g * M:(g + 1) * M], a)
        gemm_grouped([((M, H2, a[g * M:(g + 1) * M], H1, net[4].weight, H1, H1, f[g * M:(g + 1) * M], H2),
                       dict(bias=_f32c(net[4].bias), compute=X3)) for g, net in enumerate(nets)])
        return [f[g * M:(g + 1) * M] for g in range(G)]
    cmp = _compute_for(x)
    h1 = torch.empty((G * M, H1), dtype=x.dtype, device=x.device)
    a = torch.empty_like(h1)
    f = torch.empty((G * M, H2), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    mean = torch.empty(G * M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    gemm_grouped([((M, H1, x, H, weight_operand(net[1].weight, x.dtype), H, H, h1[g * M:(g + 1) * M], H1),
                   dict(bias=_f32c(net[1].bias), compute=cmp)) for g, net in enumerate(nets)])
    lw, lb = [_f32c(net[2].weight) for net in nets], [_f32c(net[2].bias) for net in nets]
    row_ptr = (C.c_int32 * (G + 1))(*[g * M for g in range(G + 1)])
    _ck(lib.egk_rowln_group_fwd(_stream(), _p(h1), _ptr_array(lw), _ptr_array(lb), row_ptr, G, _p(a), _p(mean), _p(rstd), H1,
                                float(nets[0][2].eps), 1, _dt(h1)), "egk_rowln_group_fwd")
    gemm_grouped([((M, H2, a[g * M:(g + 1) * M], H1, weight_operand(net[4].weight, x.dtype), H1, H1, f[g * M:(g + 1) * M], H2),
                   dict(bias=_f32c(net[4].bias), compute=cmp)) for g, net in enumerate(nets)])
    return [f[g * M:(g + 1) * M] for g in range(G)]


def grouped_projection_ok(xs, nets, specs=None) -> bool:
    """Whether ``grouped_projection`` can serve these task batches: bf16 activations, 2 .. 4 standard projection heads
    (Dropout(0 / eval) -> Linear -> LayerNorm -> ReLU -> Linear) of equal widths, every width a multiple of 64 (whole K
    tiles of the pipelined contraction), every batch a multiple of 64 rows (the K axis of its weight gradient), and the
    parameters already living in the optimizer's flat buffers (in-place gradient slots, bf16 weight shadows)."""
    if not (2 <= len(xs) <= 4) or any(x.dtype != torch.bfloat16 or not x.is_cuda or x.dim() != 2 for x in xs):
        return False
    dims = None
    specs = specs or [None] * len(xs)
    for x, net, sp in zip(xs, nets, specs):
        if len(net) != 5 or (net[0].p > 0 and net[0].training):
            return False
        l1, ln, l2 = net[1], net[2], net[4]
        d = (l1.in_features, l1.out_features, l2.out_features)
        if dims is None:
            dims = d
        n_rows = x.shape[0] if sp is None else sp[2]
        if sp is not None and not (sp[0] >= 0 and sp[1] >= 1 and sp[0] + (sp[2] - 1) * sp[1] < x.shape[0] and sp[3].numel() == x.shape[0]):
            return False
        if d != dims or l2.in_features != d[1] or x.shape[1] != d[0] or any(v % 64 for v in d) or n_rows % 64 or n_rows == 0:
            return False
        if d[1] > 4096 or l1.bias is None or l2.bias is None:
            return False
        for p in (l1.weight, l1.bias, ln.weight, ln.bias, l2.weight, l2.bias):
            if _grad_slot(p) is None:
                return False
        if getattr(l1.weight, "_egk_shadow", None) is None or getattr(l2.weight, "_egk_shadow", None) is None:
            return False
    return True


def grouped_projection(xs, nets, compute=None, specs=None):
    """f_g = net_g(x_g) for the projection heads ``nets`` (ProjectionTask.net) of several task batches: see
    ``_GroupedProjection``.  Call ``grouped_projection_ok`` first.  ``specs[g]`` = (first, step, count, inv int64 [rows of x_g]) or
    None: head g runs on rows first + i * step (i < count) of x_g only -- the labelled rows of a task that labels one node per
    sequence (data.live_label_rows) -- and returns [count, H]; the gradient of x_g has zero rows elsewhere."""
    args = []
    for net in nets:
        l1, ln, l2 = net[1], net[2], net[4]
        args += [l1.weight, l1.bias, ln.weight, ln.bias, l2.weight, l2.bias]
    specs = tuple(specs) if specs is not None else (None,) * len(xs)
    return _GroupedProjection.apply(len(xs), _compute_for(xs[0]) if compute is None else compute, float(nets[0][2].eps), specs, *xs, *args)


# ---- GraphONE: the stages of several auxiliary tasks as ONE chain of grouped launches --------------------------------------
_g1_hook = {"fn": None}


def set_graphone_backward_hook(fn) -> None:
    """``fn()`` is called on the backward stream at the end of the grouped GraphONE backward (``_GraphOneStages``), when every
    gradient of the stage parameters is either issued or parked -- the engine starts the optimizer on that region there."""
    _g1_hook["fn"] = fn


class _GraphOneStages(torch.autograd.Function):
    """The D stages of G auxiliary tasks' GraphONE interaction (reference models/graphONE/graphONE.py:94-115, the N feature
    rows only -- see models/graphONE/graphONE.py here) as one chain of launches over all tasks: per stage G max aggregations,
    ONE grouped two-source contraction, ONE grouped row LayerNorm + ReLU, ONE grouped contraction (+ bias + residual) forward;
    backward per stage one grouped dX contraction, one grouped LayerNorm backward, one grouped launch of the 2 G dX
    contractions of the SAGE layer (the residual's gradient added in the epilogue) and ONE aggregation backward over all tasks;
    the 3 G weight gradients of a stage are parked for grouped launches -- instead of G chains of 4 D forward / ~12 D backward
    launches on G streams, each a contraction over a third of the rows.
    Inputs: f0 [G * N, H] (task g = rows g N .. (g + 1) N), per task a frozen f32 bank and nn [N, k]; per (stage, task) the
    parameters (Wl, Wr, ln_w, ln_b, W3, b3).  Parameter gradients accumulate in place (flat gradient slots)."""

    @staticmethod
    def forward(ctx, f0, G, D, N, residual, eps, compute, banks, nns, *params):
        lib = _lib.load()
        f0 = _c(f0)
        dt, dev = f0.dtype, f0.device
        H = f0.shape[1]
        P = [[params[(s * G + g) * 6:(s * G + g) * 6 + 6] for g in range(G)] for s in range(D)]
        H1 = P[0][0][0].shape[0]
        k = nns[0].shape[1]
        row_ptr = (C.c_int32 * (G + 1))(*[g * N for g in range(G + 1)])
        sl = lambda t, g: t[g * N:(g + 1) * N]
        saved, ops_w = [], []
        f = f0
        for s in range(D):
            m = torch.empty_like(f)
            arg = torch.empty((G * N, H), dtype=torch.uint8, device=dev)
            _ck(lib.egk_gather_max_group_fwd(_stream(), _p(f), _ptr_array(banks), _ptr_array(nns), G, _p(m), _p(arg), N, H, k, _dt(f)),
                "egk_gather_max_group_fwd")
            Wl = [weight_operand(P[s][g][0], dt) for g in range(G)]
            Wr = [weight_operand(P[s][g][1], dt) for g in range(G)]
            W3 = [weight_operand(P[s][g][4], dt) for g in range(G)]
            h = torch.empty((G * N, H1), dtype=dt, device=dev)
            gemm_grouped([((N, H1, sl(m, g), H, Wl[g], H, H, sl(h, g), H1),
                           dict(A2=sl(f, g), lda2=H, B2=Wr[g], ldb2=H, K2=H, compute=compute)) for g in range(G)])
            a = torch.empty_like(h)
            mean = torch.empty(G * N, dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            lw, lb = [_f32c(P[s][g][2]) for g in range(G)], [_f32c(P[s][g][3]) for g in range(G)]
            _ck(lib.egk_rowln_group_fwd(_stream(), _p(h), _ptr_array(lw), _ptr_array(lb), row_ptr, G, _p(a), _p(mean), _p(rstd), H1,
                                        eps, 1, _dt(h)), "egk_rowln_group_fwd")
            out = torch.empty((G * N, H), dtype=dt, device=dev)
            gemm_grouped([((N, H, sl(a, g), H1, W3[g], H1, H1, sl(out, g), H),
                           dict(bias=_f32c(P[s][g][5]), residual=sl(f, g) if residual else None, ldr=H, compute=compute))
                          for g in range(G)])
            saved += [f, m, arg, h, a, mean, rstd]
            ops_w.append((Wl, Wr, W3, lw, lb))
            f = out
        ctx.dims, ctx.residual, ctx.compute, ctx.k = (G, D, N, H, H1), residual, compute, k
        ctx.P, ctx.ops_w = P, ops_w
        ctx.save_for_backward(*saved)
        return tuple(sl(f, g) for g in range(G))

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.load()
        G, D, N, H, H1 = ctx.dims
        saved, P, cmp = ctx.saved_tensors, ctx.P, ctx.compute
        dt, dev = saved[0].dtype, saved[0].device
        sl = lambda t, g: t[g * N:(g + 1) * N]
        row_ptr = (C.c_int32 * (G + 1))(*[g * N for g in range(G + 1)])
        dy = [_operand_rows(d if d is not None else torch.zeros((N, H), dtype=dt, device=dev), dt) for d in dys]
        slots = [[[_grad_slot(p) for p in P[s][g]] for g in range(G)] for s in range(D)]
        if any(x is None for ss in slots for sg in ss for x in sg):
            raise RuntimeError("graphone_stages: the parameters need in-place gradient slots (optim.FlatAdam materialised)")
        need_f0 = ctx.needs_input_grad[0]
        a0, m0 = saved[4], saved[1]
        park = (_wgrad_groupable(H, H1, dy[0], dy[0].stride(0), a0, H1, N, cmp) and _wgrad_groupable(H1, H, a0, H1, m0, H, N, cmp)
                and all(d.stride(0) % 8 == 0 and d.data_ptr() % 16 == 0 for d in dy))
        pending, reds = [], []  # without the parking queue: grouped launches of <= 8 on the side stream, below

        def wgrad(pa, pk, keep):
            if not (park and _wgrad_defer(pa, pk, keep, park_on_excluded=True)):
                pending.append((pa, pk, keep))
        df = None
        for s in reversed(range(D)):
            f, m, arg, h, a, mean, rstd = saved[7 * s:7 * s + 7]
            Wl, Wr, W3, lw, lb = ctx.ops_w[s]
            da = torch.empty_like(a)
            stamp("graphone_bwd_stage", seq=True)
            gemm_grouped([((N, H1, dy[g], dy[g].stride(0), W3[g], H1, H, sl(da, g), H1), dict(transB=True, compute=cmp)) for g in range(G)])
            for g in range(G):
                wgrad((H, H1, dy[g], dy[g].stride(0), sl(a, g), H1, N, slots[s][g][4], H1),
                      dict(transA=True, transB=True, accumulate=True, compute=cmp, dbias=slots[s][g][5]), (dy[g], a))
            dh = torch.empty_like(h)
            grid = lib.egk_rowln_bwd_ws_rows(N)
            ws = torch.empty(G * grid * 2 * H1 * 4, dtype=torch.uint8, device=dev)
            _ck(lib.egk_rowln_group_bwd(_stream(), _p(da), _p(h), _ptr_array(lw), _ptr_array(lb), row_ptr, G, _p(mean), _p(rstd),
                                        _p(dh), _p(ws), H1, 1, _dt(h)), "egk_rowln_group_bwd")
            for g in range(G):
                red = (ws[g * grid * 2 * H1 * 4:], slots[s][g][2], slots[s][g][3], N, H1, 0)
                if park:
                    _wgrad_defer_reduce(*red)
                else:
                    reds.append(red)
            if s > 0 or need_f0:
                dm = torch.empty((G * N, H), dtype=dt, device=dev)
                df = torch.empty_like(dm)
                gemm_grouped([((N, H, sl(dh, g), H1, Wl[g], H, H1, sl(dm, g), H), dict(transB=True, compute=cmp)) for g in range(G)]
                             + [((N, H, sl(dh, g), H1, Wr[g], H, H1, sl(df, g), H),
                                 dict(transB=True, compute=cmp, residual=dy[g] if ctx.residual else None, ldr=dy[g].stride(0)))
                                for g in range(G)])
                _ck(lib.egk_gather_max_bwd(_stream(), _p(dm), _p(arg), _p(df), G * N, H, ctx.k, 1, _dt(dm)), "egk_gather_max_bwd")
            for g in range(G):
                wgrad((H1, H, sl(dh, g), H1, sl(m, g), H, N, slots[s][g][0], H),
                      dict(transA=True, transB=True, accumulate=True, compute=cmp), (dh, m))
                wgrad((H1, H, sl(dh, g), H1, sl(f, g), H, N, slots[s][g][1], H),
                      dict(transA=True, transB=True, accumulate=True, compute=cmp), (dh, f))
            if s > 0:
                dy = [sl(df, g) for g in range(G)]
        if park and _g1_hook["fn"] is not None:
            _g1_hook["fn"]()  # (every gradient of the interaction's parameters has been issued or parked: engine.EgoPackStep)
        for i in range(0, len(pending), 8):
            chunk = pending[i:i + 8]
            keep = tuple(t for _, _, kp in chunk for t in kp)
            _wgrad_launch(True, keep, lambda chunk=chunk: gemm_grouped([(pa, pk) for pa, pk, _ in chunk]))
        if reds:
            _wgrad_launch(True, tuple(r[0] for r in reds), lambda: _launch_reductions(reds))
        return (df if need_f0 else None, *([None] * (8 + 6 * G * D)))


def graphone_stages_ok(G: int, N: int, H: int, banks, stage_lists, freeze: bool) -> bool:
    """Whether ``graphone_stages`` can serve an interaction of G tasks with N feature rows of width H each: 2 .. 4 tasks, bf16
    activations, N a multiple of 64 (the K axis of the weight gradients), widths in multiples of 64, frozen banks, and every
    stage parameter living in the optimizer's flat buffers (in-place gradient slots)."""
    if not (2 <= G <= 4) or not freeze or _state["act"] != torch.bfloat16 or not switches.enabled("graphone_grouped"):
        return False
    if N <= 0 or N % 64 or H % 64 or any(b.requires_grad or b.dim() != 2 or b.shape[1] != H or not b.is_cuda for b in banks):
        return False
    D = len(stage_lists[0])
    H1 = stage_lists[0][0].module_0.lin_l.weight.shape[0]
    if D == 0 or H1 % 64 or H1 > 4096:
        return False
    for stages in stage_lists:
        if len(stages) != D:
            return False
        for st in stages:
            c, ln, l3 = st.module_0, st.module_1, st.module_3
            if (c.lin_l.weight.shape != (H1, H) or c.lin_r.weight.shape != (H1, H) or c.lin_l.bias is not None
                    or l3.weight.shape != (H, H1) or l3.bias is None or ln.weight.shape != (H1,)):
                return False
            for p in (c.lin_l.weight, c.lin_r.weight, ln.weight, ln.bias, l3.weight, l3.bias):
                if _grad_slot(p) is None or not p.requires_grad:
                    return False
    return True


def to_act_rows(feats):
    """[sum rows, H] activation-type copy of several row blocks of one width, one below the other: ONE conversion launch when the
    blocks already are consecutive slices of one buffer (the grouped auxiliary projections), one per block otherwise."""
    want = _state["act"]
    f0 = feats[0]
    base = f0._base
    H = f0.shape[1]
    rows = [f.shape[0] for f in feats]
    if base is not None and base.dim() == 2 and base.is_contiguous() and base.shape == (sum(rows), H):
        off, whole = base.data_ptr(), True
        for f in feats:
            whole = whole and f._base is base and f.is_contiguous() and f.data_ptr() == off
            off += f.shape[0] * H * base.element_size()
        if whole:
            return base if base.dtype == want else cast_raw(base, want)
    out = torch.empty((sum(rows), H), dtype=want, device=f0.device)
    r0 = 0
    for f in feats:
        f = _rm(f)
        _ck(_lib.load().egk_cast_rows(_stream(), _p(f), _dt(f), f.stride(0), _p(out[r0:]), _dt(out), H, f.shape[0], H, 0), "egk_cast_rows")
        r0 += f.shape[0]
    return out


def graphone_stages(f0, banks, nns, stage_lists, residual: bool):
    """[f_g after D stages] for the G tasks' features f0 [G * N, H] (activation type; see ``_GraphOneStages``; ask
    ``graphone_stages_ok`` first)."""
    G, D = len(banks), len(stage_lists[0])
    N = f0.shape[0] // G
    params = []
    for s in range(D):
        for g in range(G):
            st = stage_lists[g][s]
            params += [st.module_0.lin_l.weight, st.module_0.lin_r.weight, st.module_1.weight, st.module_1.bias, st.module_3.weight,
                       st.module_3.bias]
    eps = float(stage_lists[0][0].module_1.eps)
    return list(_GraphOneStages.apply(f0, G, D, N, bool(residual), eps, _compute_for(f0), [_f32c(b.detach()) for b in banks],
                                      [n.contiguous() for n in nns], *params))


# ---- row LayerNorm (+ReLU, +dropout) ------------------------------------------------------------------
class _RowLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, relu, p, training):
        _need_gpu(x, w)
        lib = _lib.load()
        slabbed = getattr(x, "_egk_slabs", None) is not None and _dual["replay"] is None
        if not slabbed:
            x = _c(x)
        rows, cols = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        p_eff = float(p) if training else 0.0
        mask = torch.empty((rows, cols), dtype=torch.uint8, device=x.device) if p_eff > 0 else None
        seed, off = _next_rng(rows * max(cols, 4096)) if p_eff > 0 else (0, 0)
        wc, bc = _f32c(w), _f32c(b)
        dev_off = _p(rng_device_offset(x.device)) if p_eff > 0 else None
        taped = _tape_take("rowln", x) if p_eff == 0 else None
        if taped is not None:
            y, mean, rstd = _r16(taped, "y"), taped["mean"], taped["rstd"]
        else:
            tee = _tee_arm(y)
            xin, took = _slab_consumer(x) if slabbed else (None, False)  # (the reduce of the contraction that made x rides here)
            _ck(lib.egk_rowln_fwd(_stream(), xin if took else _p(x), _p(wc), _p(bc), _p(y), _p(mean), _p(rstd), _p(mask), rows, cols, eps,
                                  int(relu), p_eff, seed, off, dev_off, _dt(x)),
                "egk_rowln_fwd")
            _tee_done(y, tee)
            if took:
                _slabs_written(x)
            if p_eff == 0:
                _tape_put("rowln", y=y, mean=mean, rstd=rstd)
        ctx.relu, ctx.p = relu, p_eff
        ctx.params = (w, b)
        ctx.save_for_backward(x, wc, bc, mean, rstd, mask)
        if _mask_tap["on"] and mask is not None:
            _mask_tap["masks"].append(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        stamp("bwd_rowln", seq=True)
        x, w, b, mean, rstd, mask = ctx.saved_tensors
        wp, bp = ctx.params
        rows, cols = x.shape
        dy = _match(dy, x.dtype)
        dx = torch.empty_like(x)
        slot_w, slot_b = _grad_slot(wp), _grad_slot(bp)
        dw = slot_w if slot_w is not None else torch.zeros_like(w)
        db = slot_b if slot_b is not None else torch.zeros_like(b)
        nbytes = 2 * lib.egk_rowln_bwd_ws_rows(rows) * cols * 4
        if _ln_reduce_on_side(slot_w, slot_b, x):
            # dw / db feed nothing but the optimizer: their reduction goes to the weight-gradient side stream, from a
            # workspace of its own (the shared one may be rewritten by the next launch of this stream)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            forked = _rows_fork()
            _ck(lib.egk_rowln_bwd(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(mean), _p(rstd), _p(mask), _p(dx), None, None,
                                  _p(ws), rows, cols, int(ctx.relu), ctx.p, _dt(x)), "egk_rowln_bwd")
            _rows_join(forked)
            if _wq["on"] and _wgrad["enabled"]:
                _wgrad_defer_reduce(ws, dw, db, rows, cols, 0)
            else:
                _wgrad_launch(True, (ws,), lambda: _ck(lib.egk_ln_bwd_reduce(_stream(), _p(ws), _p(dw), _p(db), rows, cols, 0),
                                                       "egk_ln_bwd_reduce"))
            return dx, None, None, None, None, None, None
        ws = workspace(nbytes, x.device)
        _ck(lib.egk_rowln_bwd(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(mean), _p(rstd), _p(mask), _p(dx), _p(dw), _p(db),
                              _p(ws), rows, cols, int(ctx.relu), ctx.p, _dt(x)), "egk_rowln_bwd")
        return dx, (None if slot_w is not None else dw), (None if slot_b is not None else db), None, None, None, None


_mask_tap = {"on": False, "masks": []}


class tap_dropout_masks:
    """``with ops.tap_dropout_masks() as masks:`` -- the keep masks (uint8, 1 = keep) of every fused LayerNorm + dropout launch
    issued inside, in issue order: tests hand them to the oracle (``oracle.path.trn_pooling(masks=...)``) so that a step with
    active dropout can be compared element by element."""

    def __enter__(self):
        self.prev = (_mask_tap["on"], _mask_tap["masks"])
        _mask_tap["on"], _mask_tap["masks"] = True, []
        return _mask_tap["masks"]

    def __exit__(self, *a):
        _mask_tap["on"], _mask_tap["masks"] = self.prev


def row_layernorm(x, w, b, eps=1e-5, relu=False, p=0.0, training=False):
    """dropout(relu(LayerNorm(x))) in one launch (nn.LayerNorm -> nn.ReLU -> nn.Dropout)."""
    return _RowLN.apply(x, w, b, float(eps), bool(relu), float(p), bool(training))


def last_rowln_mask(y: torch.Tensor):
    """Keep-mask (uint8) saved by the row_layernorm that produced ``y`` (tests feed it to the oracle)."""
    return y.grad_fn.saved_tensors[5] if y.grad_fn is not None else None


# ---- graph-mode LayerNorm + LeakyReLU -------------------------------------------------------------------
def _ln_bwd_stats_launch(lnctx, args, kw, dy_out):
    """Launch the dX contraction ``gemm(*args, **kw)`` whose result ``dy_out`` is the gradient at the output of the graph
    LayerNorm + LeakyReLU described by ``lnctx``; if the launch can, its epilogue also takes that LayerNorm's backward sums
    (egk_gemm st_mode 2) and leaves them in ``lnctx['bwd']`` for ``_GraphLN.backward``."""
    req = dict(mode=2, seg_ptr=lnctx["seg_ptr"], n_seg=lnctx["n_seg"], min_rows=lnctx["min_rows"], x=lnctx["x"],
               stats=lnctx["stats"], w=lnctx["w"], b=lnctx["b"], slope=lnctx["slope"])
    ok = (_ln_fusion["on"] and lnctx["x"].dtype == dy_out.dtype and lnctx["x"].shape == dy_out.shape)
    got = _gemm_with_stats(args, kw, req) if ok else (gemm(*args, **kw) and None)
    lnctx["bwd"] = (got[0], got[1], dy_out.data_ptr()) if got else None


_ln_fusion = {"on": True}  # development knob: False = every graph LayerNorm runs its own statistics passes

# Exact cross-rank statistics (data parallelism): the reference is one process, its graph-mode LayerNorm sees the whole
# batch (models/graph.py:43); with the batch sharded over ranks the default is per-rank statistics (every replica is the
# reference at its local batch size).  With an exchange function installed, every graph LayerNorm sums its segment
# statistics over the ranks between its statistics pass and its normalising pass, forward and backward -- the result is
# the single-process result at the GLOBAL batch (up to summation order).  ``fn(buf)`` must sum the f64 tensor ``buf``
# [n_seg, 3] over the ranks in place, ordered on the current stream (dist.GradSync.sum_small).  Collectives cannot be
# captured: the engine steps eagerly in this mode.
_ln_exchange = {"fn": None}


def set_graph_ln_exchange(fn):
    """Install (or with None remove) the cross-rank sum of the graph LayerNorm statistics; returns the previous one."""
    prev = _ln_exchange["fn"]
    _ln_exchange["fn"] = fn
    return prev


def graph_ln_exchange_on() -> bool:
    return _ln_exchange["fn"] is not None


def _exchange_segment_sums(partials, n_blocks, seg_ptr, n_seg, cols):
    """Per-block (a, b) sums per segment of THIS rank -> [1, n_seg, 2] f64 sums over all ranks, scaled by local / global
    element count (the normalising kernels divide by the local count of a segment)."""
    loc = partials.view(torch.float64)[: n_blocks * n_seg * 2].view(n_blocks, n_seg, 2).sum(0)
    cnt = (seg_ptr[1:] - seg_ptr[:-1]).to(torch.float64) * cols
    buf = torch.cat([loc, cnt[:, None]], 1).contiguous()
    _ln_exchange["fn"](buf)
    scale = torch.where(buf[:, 2] > 0, cnt / buf[:, 2].clamp_min(1.0), torch.zeros_like(cnt))
    return (buf[:, :2] * scale[:, None]).contiguous().view(1, n_seg, 2)


class _GraphLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, seg_ptr, eps, slope, partials=None, lnctx=None):
        _need_gpu(x, w, seg_ptr)
        lib = _lib.load()
        slabbed = (getattr(x, "_egk_slabs", None) is not None and _dual["replay"] is None and partials is None
                   and _ln_exchange["fn"] is None)
        if not slabbed:
            x = _c(x)
        rows, cols = x.shape
        n_seg = seg_ptr.numel() - 1
        y = torch.empty_like(x)
        stats = torch.empty(n_seg * 2, dtype=torch.float32, device=x.device)
        wc, bc = _f32c(w), _f32c(b)
        taped = _tape_take("graphln", x) if _ln_exchange["fn"] is None else None
        if taped is not None:
            y, stats = _r16(taped, "y"), taped["stats"]
            if stats.numel() != n_seg * 2:
                raise RuntimeError("dual_replay: taped graph LayerNorm statistics have another segment count")
        elif _ln_exchange["fn"] is not None:  # exact cross-rank statistics: local sums -> sum over ranks -> normalise
            nb = lib.egk_graphln_stats_blocks(rows)
            loc = torch.empty(nb * n_seg * 2, dtype=torch.float64, device=x.device)
            _ck(lib.egk_graphln_stats(_stream(), _p(x), _p(seg_ptr), n_seg, rows, cols, _p(loc), _dt(x)), "egk_graphln_stats")
            glob = _exchange_segment_sums(loc, nb, seg_ptr, n_seg, cols)
            _ck(lib.egk_graphln_fwd_apply(_stream(), _p(x), _p(wc), _p(bc), _p(y), _p(stats), _p(seg_ptr), n_seg, rows, cols, eps,
                                          slope, _p(glob), 1, _dt(x)), "egk_graphln_fwd_apply")
            lnctx = None
        elif partials is not None:  # the segment sums came with the contraction that produced x
            tee = _tee_arm(y)
            _ck(lib.egk_graphln_fwd_apply(_stream(), _p(x), _p(wc), _p(bc), _p(y), _p(stats), _p(seg_ptr), n_seg, rows, cols, eps,
                                          slope, _p(partials[0]), partials[1], _dt(x)), "egk_graphln_fwd_apply")
            _tee_done(y, tee)
        else:
            ws = workspace(lib.egk_graphln_ws_bytes(rows, cols, n_seg), x.device)
            tee = _tee_arm(y)
            xin, took = _slab_consumer(x) if slabbed else (None, False)  # (the statistics launch reduces the slabs on its way)
            _ck(lib.egk_graphln_fwd(_stream(), xin if took else _p(x), _p(wc), _p(bc), _p(y), _p(stats), _p(seg_ptr), n_seg, rows, cols, eps,
                                    slope, _p(ws), _dt(x)), "egk_graphln_fwd")
            _tee_done(y, tee)
            if took:
                _slabs_written(x)
        if taped is None and _ln_exchange["fn"] is None:
            _tape_put("graphln", y=y, stats=stats)
        ctx.eps, ctx.slope = eps, slope
        ctx.params = (w, b)
        ctx.lnctx = lnctx
        ctx.exchange = _ln_exchange["fn"] is not None
        if lnctx is not None:  # what the consumer's dX epilogue needs to take this layer's backward sums
            lnctx.update(x=x, stats=stats, w=wc, b=bc, slope=slope, seg_ptr=seg_ptr, n_seg=n_seg, bwd=None)
        ctx.save_for_backward(x, wc, bc, stats, seg_ptr)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        stamp("bwd_graphln", seq=True)
        x, w, b, stats, seg_ptr = ctx.saved_tensors
        wp, bp = ctx.params
        rows, cols = x.shape
        n_seg = seg_ptr.numel() - 1
        dy = _match(dy, x.dtype)
        dx = torch.empty_like(x)
        slot_w, slot_b = _grad_slot(wp), _grad_slot(bp)
        dw = slot_w if slot_w is not None else torch.zeros_like(w)
        db = slot_b if slot_b is not None else torch.zeros_like(b)
        if ctx.exchange:  # exact cross-rank statistics (forward ran with an exchange function)
            if _ln_exchange["fn"] is None:
                raise RuntimeError("graph LayerNorm: forward summed its statistics over the ranks, backward has no exchange function")
            ws = torch.empty(lib.egk_graphln_ws_bytes(rows, cols, n_seg), dtype=torch.uint8, device=x.device)
            _ck(lib.egk_graphln_bwd_stats(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(stats), _p(seg_ptr), n_seg, rows, cols, ctx.slope,
                                          _p(ws), _dt(x)), "egk_graphln_bwd_stats")
            nb = lib.egk_graphln_stats_blocks(rows)
            glob = _exchange_segment_sums(ws[: nb * n_seg * 16], nb, seg_ptr, n_seg, cols)
            _ck(lib.egk_graphln_bwd_finish(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(stats), _p(dx), _p(dw), _p(db), _p(seg_ptr),
                                           n_seg, rows, cols, ctx.eps, ctx.slope, _p(glob), 1, _p(ws), _dt(x)), "egk_graphln_bwd_finish")
            return dx, (None if slot_w is not None else dw), (None if slot_b is not None else db), None, None, None, None, None
        pre = ctx.lnctx.get("bwd") if ctx.lnctx is not None else None
        if ctx.lnctx is not None:
            ctx.lnctx["bwd"] = None
        if pre is not None and pre[2] == dy.data_ptr():
            # the segment sums came with the contraction that produced dy: ONE pass (dx + the dw / db partial rows)
            ws_col = torch.empty(lib.egk_rowln_bwd_ws_rows(rows) * 2 * cols, dtype=torch.float32, device=x.device)
            forked = _rows_fork()
            _ck(lib.egk_graphln_bwd_apply(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(stats), _p(dx), _p(seg_ptr), n_seg, rows,
                                          cols, ctx.eps, ctx.slope, _p(pre[0]), pre[1], _p(ws_col), _dt(x)), "egk_graphln_bwd_apply")
            _rows_join(forked)
            if _ln_reduce_on_side(slot_w, slot_b, x):
                if _wq["on"] and _wgrad["enabled"]:
                    _wgrad_defer_reduce(ws_col, dw, db, rows, cols, 0)
                else:
                    _wgrad_launch(True, (ws_col,), lambda: _ck(lib.egk_ln_bwd_reduce(_stream(), _p(ws_col), _p(dw), _p(db), rows, cols, 0),
                                                               "egk_ln_bwd_reduce"))
                return dx, None, None, None, None, None, None, None
            _ck(lib.egk_ln_bwd_reduce(_stream(), _p(ws_col), _p(dw), _p(db), rows, cols, 0), "egk_ln_bwd_reduce")
            return dx, (None if slot_w is not None else dw), (None if slot_b is not None else db), None, None, None, None, None
        nbytes = lib.egk_graphln_ws_bytes(rows, cols, n_seg)
        if _ln_reduce_on_side(slot_w, slot_b, x):  # (see _RowLN.backward)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            _ck(lib.egk_graphln_bwd(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(stats), _p(dx), None, None, _p(seg_ptr),
                                    n_seg, rows, cols, ctx.eps, ctx.slope, _p(ws), _dt(x)), "egk_graphln_bwd")
            if _wq["on"] and _wgrad["enabled"]:
                _wgrad_defer_reduce(ws, dw, db, rows, cols, n_seg)
            else:
                _wgrad_launch(True, (ws,), lambda: _ck(lib.egk_ln_bwd_reduce(_stream(), _p(ws), _p(dw), _p(db), rows, cols, n_seg),
                                                       "egk_ln_bwd_reduce"))
            return dx, None, None, None, None, None, None, None
        ws = workspace(nbytes, x.device)
        _ck(lib.egk_graphln_bwd(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(stats), _p(dx), _p(dw), _p(db), _p(seg_ptr),
                                n_seg, rows, cols, ctx.eps, ctx.slope, _p(ws), _dt(x)), "egk_graphln_bwd")
        return dx, (None if slot_w is not None else dw), (None if slot_b is not None else db), None, None, None, None, None


def graph_layernorm_lrelu(x, w, b, seg_ptr, eps=1e-5, slope=0.2, partials=None, min_seg_rows=0, return_ctx=False):
    """LeakyReLU(gnn.LayerNorm(mode='graph')(x)) with statistics per row segment (int32 seg_ptr).
    ``partials``: (double [blocks][n_seg][2], blocks) per-block (sum, sum of squares) per segment when the contraction that
    produced x took them in its epilogue.  ``return_ctx``: also return the context a consumer (``sage_mean_layer`` /
    ``linear`` with ``ln_in=``) needs to take this layer's BACKWARD sums in its dX epilogue (``min_seg_rows``: the
    shortest segment, a host integer)."""
    lnctx = {"min_rows": int(min_seg_rows)} if (return_ctx and min_seg_rows > 0) else None
    if _ln_exchange["fn"] is not None:  # statistics summed over the ranks: nothing rides on a neighbouring contraction
        lnctx = partials = None
    y = _GraphLN.apply(x, w, b, seg_ptr, float(eps), float(slope), partials, lnctx)
    return (y, lnctx) if return_ctx else y


# ---- positional encoding add --------------------------------------------------------------------------
_pe_tables = {}


def _pe_freq_id(freq):
    """Identity of a frequency buffer by CONTENT (models built alike share their tables; a table is never freed, so a
    captured graph that reads one stays valid): a host hash taken the first time a buffer is seen -- outside graph captures
    only (it synchronises); None while capturing an unseen buffer.  The id is kept ON the tensor, with the version counter it
    was taken at (a raw address may be handed to another buffer once this one is freed)."""
    c = getattr(freq, "_egk_pe_id", None)
    if c is not None and c[0] == freq._version and c[1] == freq.data_ptr() and c[2] == freq.numel():
        return c[3]
    if torch.cuda.is_current_stream_capturing():
        return None
    fid = hash(freq.detach().float().cpu().numpy().tobytes())
    try:
        freq._egk_pe_id = (freq._version, freq.data_ptr(), freq.numel(), fid)
    except Exception:  # noqa: BLE001  (a tensor type that takes no attributes: hashed again next time)
        pass
    return fid


def _pe_table(freq, fid, pos_min: int, n_pos: int, cols: int):
    """[n_pos, cols] f32 table of PE(p), p in [pos_min, pos_min + n_pos), cached per (frequency content, range, device) for
    the life of the process (128 KB for 32 positions x 1024 channels).  Built outside graph captures; None if it would have
    to be built inside one."""
    key = (fid, int(pos_min), int(n_pos), int(cols), freq.device.index)
    t = _pe_tables.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        t = torch.empty((n_pos, cols), dtype=torch.float32, device=freq.device)
        _ck(_lib.load().egk_pe_table(_stream(), _p(_f32c(freq)), int(pos_min), int(n_pos), int(cols), _p(t)), "egk_pe_table")
        torch.cuda.current_stream().synchronize()  # (built once; later readers may sit on any stream)
        _pe_tables[key] = t
    return t


class _PEAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, freq, pos_range):
        _need_gpu(x, pos, freq)
        slabbed = getattr(x, "_egk_slabs", None) is not None and _dual["replay"] is None
        if not slabbed:
            x = _c(x)
        rows, cols = x.shape
        y = torch.empty_like(x)
        lib = _lib.load()
        n_pos = (pos_range[1] - pos_range[0] + 1) if pos_range is not None else 0
        table = None
        if 0 < n_pos <= 4096:
            fid = _pe_freq_id(freq)
            table = _pe_table(freq, fid, pos_range[0], n_pos, cols) if fid is not None else None
        posc, fr = pos.contiguous(), _f32c(freq)
        taped = _tape_take("pe_add", x)
        if taped is not None:
            return _r16(taped, "y")
        if slabbed and table is None:
            x = _c(x)  # (the direct evaluation takes no slabs: reduce first)
            slabbed = False
        tee = _tee_arm(y)
        if table is not None:
            xin, took = _slab_consumer(x) if slabbed else (None, False)
            _ck(lib.egk_pe_add_table(_stream(), xin if took else _p(x), _p(posc), _p(fr), _p(table), int(pos_range[0]), int(n_pos),
                                     _p(y), rows, cols, _dt(x)), "egk_pe_add_table")
            if took:
                _slabs_written(x)
        else:
            _ck(lib.egk_pe_add(_stream(), _p(x), _p(posc), _p(fr), _p(y), rows, cols, _dt(x)), "egk_pe_add")
        _tee_done(y, tee)
        _tape_put("pe_add", y=y)
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None


def pe_add(x, pos, freq, pos_range=None):
    """x + PositionalEncoding(pos).  ``pos_range`` = (min, max) of the integer positions when the host knows it (collated
    batches carry it): PE is then evaluated once per distinct position (a cached [max - min + 1, C] table) instead of once
    per node -- same bits."""
    if pos.dtype != torch.int64:
        pos = pos.to(torch.int64)
    return _PEAdd.apply(x, pos, freq, pos_range)


# ---- CSR mean aggregation -------------------------------------------------------------------------------
def _csr_gather(x, rowptr, col, wgt, gate, out, heavy=None, heavy_mode=0, band=None):
    """One ``egk_csr_gather`` call; ``heavy`` = ascending int32 ids of the rows with more than
    ``egk_csr_heavy_threshold()`` edges (data.build_csr lists them), or None / empty; ``heavy_mode`` 1 when none of them
    has more than data.HEAVY_IN_LAUNCH_DEGREE edges (summed inside the launch), 0 for the split launches."""
    lib = _lib.load()
    rows, cols = x.shape
    nh = int(heavy.numel()) if heavy is not None else 0
    ws = workspace(lib.egk_csr_heavy_ws_bytes(nh, cols), x.device) if nh and not heavy_mode else None
    # (the split tee: only when every row is finished inside the gather launch itself)
    tee = _tee_arm(out) if (wgt is None and gate is None and (nh == 0 or heavy_mode)) else None
    if band is not None and wgt is None and gate is None and _banded["on"]:
        # forward mean aggregation: rows whose neighbours are {i - 1, i, i + 1} need no index fetch (data.band_codes)
        _ck(lib.egk_csr_gather_banded(_stream(), _p(x), _p(rowptr), _p(col), _p(band), _p(out), rows, cols, _dt(x),
                                      _p(heavy) if nh else None, nh, _p(ws) if ws is not None else None, int(heavy_mode)),
            "egk_csr_gather_banded")
        _tee_done(out, tee)
        return
    _ck(lib.egk_csr_gather(_stream(), _p(x), _p(rowptr), _p(col), _p(wgt), _p(gate), _p(out), rows, cols, _dt(x),
                           _p(heavy) if nh else None, nh, _p(ws) if ws is not None else None, int(heavy_mode)), "egk_csr_gather")
    _tee_done(out, tee)


_banded = {"on": switches.enabled("banded_gather")}  # development knob


class _CSRMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rowptr, col, t_rowptr, t_col, t_wgt, heavy, t_heavy, heavy_mode=0, t_heavy_mode=0, band=None):
        _need_gpu(x, rowptr, col)
        x = _c(x)
        out = torch.empty_like(x)
        _csr_gather(x, rowptr, col, None, None, out, heavy, heavy_mode, band)
        ctx.save_for_backward(t_rowptr, t_col, t_wgt, t_heavy)
        ctx.t_heavy_mode = t_heavy_mode
        return out

    @staticmethod
    def backward(ctx, dout):
        t_rowptr, t_col, t_wgt, t_heavy = ctx.saved_tensors
        dout = _c(dout)
        dx = torch.empty_like(dout)
        _csr_gather(dout, t_rowptr, t_col, t_wgt, None, dx, t_heavy, ctx.t_heavy_mode)
        return dx, None, None, None, None, None, None, None, None, None, None


def csr_mean_aggregate(x, graph):
    """agg[i] = mean_{j->i} x[j] (0 without in-edges); ``graph`` = egopack_amd.data.CSRGraph."""
    return _CSRMean.apply(x, graph.rowptr, graph.col, graph.t_rowptr, graph.t_col, graph.t_wgt,
                          getattr(graph, "heavy", None), getattr(graph, "t_heavy", None), getattr(graph, "heavy_mode", 0),
                          getattr(graph, "t_heavy_mode", 0), getattr(graph, "band", None))


class _SageMean(torch.autograd.Function):
    """One SAGEConv(project=True, aggr='mean') layer as ONE autograd node (reference models/graph.py:42):
        xp  = relu(h @ Wp.T + bp);  agg = mean_{j->i} xp_j;  out = agg @ Wl.T + h @ Wr.T + bl
    forward 3 launches; backward 6: the ReLU gate rides on the transposed gather, d_h = d_out @ Wr + d_pre @ Wp is one
    two-source contraction (no gradient-accumulation add), the two bias gradients ride on their dW launches."""

    @staticmethod
    def forward(ctx, h, Wp, bp, Wl, bl, Wr, rowptr, col, t_rowptr, t_col, t_wgt, compute, heavy=None, t_heavy=None,
                heavy_mode=0, t_heavy_mode=0, ln_out=None, ln_in=None, res_src=None, band=None):
        _need_gpu(h, Wp, Wl, Wr)
        lib = _lib.load()
        h = _c(h)
        N, H = h.shape
        dt = h.dtype
        Wp_o, Wl_o, Wr_o = weight_operand(Wp, dt), weight_operand(Wl, dt), weight_operand(Wr, dt)
        xp = torch.empty_like(h)
        agg = torch.empty_like(h)
        p_args, p_kw = (N, H, h, H, Wp_o, H, H, xp, H), dict(bias=_f32c(bp), act=1, compute=compute)
        taped = _tape_take("sage_mean", h)
        Ho = Wl.shape[0]
        if taped is not None:
            xp, agg, out = _r16(taped, "xp"), _r16(taped, "agg"), _r16(taped, "out")
            if tuple(out.shape) != (N, Ho):
                raise RuntimeError("dual_replay: a taped SAGE layer result has another shape")
            if ln_out is not None:
                ln_out["partials"] = None  # (the graph LayerNorm that follows takes the taped statistics)
            ctx.t_heavy_mode = t_heavy_mode
        else:
            gemm(*p_args, **p_kw)
            _csr_gather(xp, rowptr, col, None, None, agg, heavy, heavy_mode, band)
            ctx.t_heavy_mode = t_heavy_mode
            out = torch.empty((N, Ho), dtype=dt, device=h.device)
            c_args = (N, Ho, agg, H, Wl_o, H, H, out, Ho)
            c_kw = dict(A2=h, lda2=H, B2=Wr_o, ldb2=H, K2=H, bias=_f32c(bl), compute=compute)
            if ln_out is not None and _ln_fusion["on"]:
                # the graph LayerNorm that follows needs (sum, sum of squares) per row segment of ``out``: taken in this epilogue
                ln_out["partials"] = _gemm_with_stats(c_args, c_kw, dict(mode=1, seg_ptr=ln_out["seg_ptr"], n_seg=ln_out["n_seg"],
                                                                         min_rows=ln_out["min_rows"]))
            else:
                _gemm_deferrable(c_args, c_kw, slab_ok=ln_out is not None)  # (``ln_out``: a graph LayerNorm is what reads ``out``)
            _tape_put("sage_mean", xp=xp, agg=agg, out=out)
        ctx.ln_in, ctx.res_src = ln_in, res_src
        ctx.compute, ctx.params = compute, (Wp, bp, Wl, bl, Wr)
        ctx.save_for_backward(h, xp, agg, Wp_o, Wl_o, Wr_o, t_rowptr, t_col, t_wgt, t_heavy)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        stamp("bwd_sage", seq=True)
        h, xp, agg, Wp_o, Wl_o, Wr_o, t_rowptr, t_col, t_wgt, t_heavy = ctx.saved_tensors
        Wp, bp, Wl, bl, Wr = ctx.params
        g = _operand_rows(d_out, h.dtype)
        N, H = h.shape
        Ho = g.shape[1]
        dev = g.device

        def slot_or_zeros(p, shape):
            s_ = _grad_slot(p)
            return (s_, None) if s_ is not None else ((z := torch.zeros(shape, dtype=torch.float32, device=dev)), z)

        dWl, rWl = slot_or_zeros(Wl, Wl_o.shape)
        dbl, rbl = slot_or_zeros(bl, (Ho,))
        dWr, rWr = slot_or_zeros(Wr, Wr_o.shape)
        dWp, rWp = slot_or_zeros(Wp, Wp_o.shape)
        dbp, rbp = slot_or_zeros(bp, (H,))
        in_place = rWl is None and rbl is None and rWr is None

        l_args, l_kw = (Ho, H, g, g.stride(0), agg, H, N, dWl, H), dict(transA=True, transB=True, accumulate=True,
                                                                        compute=ctx.compute, dbias=dbl)
        r_args, r_kw = (Ho, H, g, g.stride(0), h, H, N, dWr, H), dict(transA=True, transB=True, accumulate=True, compute=ctx.compute)

        def launch_out_grads():
            gemm(*l_args, **l_kw)
            gemm(*r_args, **r_kw)
        # eligibility is decided ONCE for both problems (same shapes; both operand pairs checked, and the stream): a pair of which
        # only the first was parked would be accumulated twice by the fallback below
        park = (in_place and ctx.compute in (BF16, F32) and not _on_excluded_stream()
                and _wgrad_groupable(Ho, H, g, g.stride(0), agg, H, N, ctx.compute)
                and _wgrad_groupable(Ho, H, g, g.stride(0), h, H, N, ctx.compute))
        if park:
            # one at a time: the first may flush a full group
            if not _wgrad_defer(l_args, l_kw, (g, agg)):
                raise RuntimeError("sage_mean_layer: a weight gradient announced as parkable was refused")
            if not _wgrad_defer(r_args, r_kw, (g, h)):
                raise RuntimeError("sage_mean_layer: a weight gradient announced as parkable was refused")
        else:
            _wgrad_launch(in_place, (g, agg, h), launch_out_grads)
        d_agg = torch.empty_like(h)
        d_pre = torch.empty_like(h)  # gradient at the projection's pre-activation: transposed gather gated by xp > 0
        a_args, a_kw = (N, H, g, g.stride(0), Wl_o, H, Ho, d_agg, H), dict(transB=True, compute=ctx.compute)
        gemm(*a_args, **a_kw)
        _csr_gather(d_agg, t_rowptr, t_col, t_wgt, xp, d_pre, t_heavy, ctx.t_heavy_mode)
        d_h = None
        if ctx.needs_input_grad[0]:
            d_h = torch.empty_like(h)
            h_args = (N, H, g, g.stride(0), Wr_o, H, Ho, d_h, H)
            h_kw = dict(A2=d_pre, lda2=H, B2=Wp_o, ldb2=H, K2=H, transB=True, compute=ctx.compute)
            extra = ctx.res_src.pop("dy", None) if ctx.res_src is not None else None
            if extra is not None:  # a gradient of h that another node handed over (the backbone's residual): added here
                if tuple(extra.shape) != tuple(d_h.shape):
                    raise RuntimeError("sage_mean_layer: the handed-over gradient does not have the layer input's shape")
                h_kw.update(residual=_c(extra), ldr=H)
            if ctx.ln_in is not None:  # d_h is dy of the graph LayerNorm that produced h: its segment sums ride on this launch
                _ln_bwd_stats_launch(ctx.ln_in, h_args, h_kw, d_h)
            else:
                gemm(*h_args, **h_kw)
        p_args, p_kw = (H, H, d_pre, H, h, H, N, dWp, H), dict(transA=True, transB=True, accumulate=True, compute=ctx.compute,
                                                               dbias=dbp)
        if not (rWp is None and rbp is None and ctx.compute in (BF16, F32) and _wgrad_defer(p_args, p_kw, (d_pre, h))):
            _wgrad_launch(rWp is None and rbp is None, (d_pre, h), lambda: gemm(*p_args, **p_kw))
        if ctx.res_src is not None and _last_wgrad["tail"] is not None:
            # the FIRST layer of the stack (its backward is the stack's last): what is parked goes out now, beside the temporal
            # pooling's backward chain -- the step's tail launch then holds the temporal pooling's weight gradients only
            flush_wgrad()
        return (d_h, rWp, rbp, rWl, rbl, rWr, None, None, None, None, None, None, None, None, None, None, None, None, None, None)


def sage_mean_layer(h, conv, graph, compute=None, ln_out=None, ln_in=None, res_src=None):
    """SAGEConv(project=True, mean) of ``conv`` (models.layers.SAGEConv parameters) on CSR ``graph``.
    ``ln_out`` = dict(seg_ptr, n_seg, min_rows): a graph LayerNorm over those row segments follows -- its forward sums are
    taken in the last contraction's epilogue and left in ``ln_out['partials']`` (None if that launch could not).
    ``ln_in``: context of the graph LayerNorm that produced ``h`` (see ``graph_layernorm_lrelu``).
    ``res_src``: a dict in which a later node of the forward graph leaves, in backward, a gradient that belongs to ``h``
    as well (``linear(..., res_sink=)``): it is added in this layer's dX epilogue."""
    return _SageMean.apply(h, conv.lin.weight, conv.lin.bias, conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight,
                           graph.rowptr, graph.col, graph.t_rowptr, graph.t_col, graph.t_wgt,
                           _compute_for(h) if compute is None else compute, getattr(graph, "heavy", None),
                           getattr(graph, "t_heavy", None), getattr(graph, "heavy_mode", 0), getattr(graph, "t_heavy_mode", 0),
                           ln_out, ln_in, res_src, getattr(graph, "band", None))


# ---- labelled rows only (the heads' row compaction) --------------------------------------------------------------------------
class _LiveRows(torch.autograd.Function):
    """Rows ``idx`` of x (int64 [cap]; -1 = an all-zero pad row); backward puts the gradient rows back at their places through the
    inverse map ``inv`` (int64 [N]; -1 = a row that was left out: zero gradient).  Both directions are ``egk_gather_rows``."""

    @staticmethod
    def forward(ctx, x, idx, inv):
        _need_gpu(x)
        x = _c(x)
        out = torch.empty((idx.numel(), x.shape[1]), dtype=x.dtype, device=x.device)
        _ck(_lib.load().egk_gather_rows(_stream(), _p(x), _dt(x), x.stride(0), x.shape[0], _p(idx), _p(out), _dt(out), idx.numel(),
                                        x.shape[1]), "egk_gather_rows")
        ctx.save_for_backward(inv)
        return out

    @staticmethod
    def backward(ctx, g):
        inv, = ctx.saved_tensors
        g = _c(g)
        out = torch.empty((inv.numel(), g.shape[1]), dtype=g.dtype, device=g.device)
        _ck(_lib.load().egk_gather_rows(_stream(), _p(g), _dt(g), g.stride(0), g.shape[0], _p(inv), _p(out), _dt(out), inv.numel(),
                                        g.shape[1]), "egk_gather_rows")
        return out, None, None


def live_rows(x, idx, inv):
    """The labelled rows of a task batch's features (data.live_label_rows) for its row-wise head; see ``_LiveRows``."""
    return _LiveRows.apply(x, idx, inv)


@torch.no_grad()
def expand_rows(v, inv, out=None):
    """A per-row vector of the compacted batch back at full length: out[i] = v[inv[i]], 0 where inv[i] < 0 (the loss vector of
    the reference has one element per node, zero on ignored nodes: criterion/wrapper.py:67-82)."""
    v2 = _c(v.detach().reshape(-1, 1))
    if out is None:
        out = torch.empty(inv.numel(), dtype=v2.dtype, device=v2.device)
    _ck(_lib.load().egk_gather_rows(_stream(), _p(v2), _dt(v2), 1, v2.shape[0], _p(inv), _p(out), _dt(out), inv.numel(), 1),
        "egk_gather_rows")
    return out


# ---- GraphONE gather-max ------------------------------------------------------------------------------
@torch.no_grad()
def _edges_by_prototype(nn: torch.Tensor, K: int):
    """Edge list of ``nn`` [N, k] grouped by prototype: (t_rowptr int32 [K+1], t_edge int32 [N*k]) with
    t_edge = n*k + j ascending inside a group.  Integer index work (a stable device sort): bit-exact."""
    flat = nn.reshape(-1)
    srt, perm = torch.sort(flat, stable=True)
    rowptr = torch.zeros(K + 1, dtype=torch.int32, device=nn.device)
    rowptr[1:] = torch.cumsum(torch.bincount(srt, minlength=K), 0).to(torch.int32)
    return rowptr, perm.to(torch.int32)


class _GatherMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, bank, nn):
        _need_gpu(f, bank, nn)
        f, bank = _c(f), _f32c(bank)
        rows, cols = f.shape
        k = nn.shape[1]
        m = torch.empty_like(f)
        arg = torch.empty((rows, cols), dtype=torch.uint8, device=f.device)
        nn = nn.contiguous()
        _ck(_lib.load().egk_gather_max_fwd(_stream(), _p(f), _p(bank), _p(nn), _p(m), _p(arg), rows, cols, k,
                                           _dt(f)), "egk_gather_max_fwd")
        ctx.k = k
        ctx.K = bank.shape[0]
        ctx.save_for_backward(arg, nn if ctx.needs_input_grad[1] else None)
        return m

    @staticmethod
    def backward(ctx, dm):
        arg, nn = ctx.saved_tensors
        lib = _lib.load()
        dm = _c(dm)
        rows, cols = dm.shape
        df = None
        if ctx.needs_input_grad[0]:
            df = torch.empty_like(dm)
            _ck(lib.egk_gather_max_bwd(_stream(), _p(dm), _p(arg), _p(df), rows, cols, ctx.k, 0, _dt(dm)), "egk_gather_max_bwd")
        dbank = None
        if ctx.needs_input_grad[1]:  # trainable prototypes (GraphONE(freeze=False)): rows gather the gradients they won
            t_rowptr, t_edge = _edges_by_prototype(nn, ctx.K)
            dbank = torch.zeros((ctx.K, cols), dtype=torch.float32, device=dm.device)
            _ck(lib.egk_gather_max_bank_grad(_stream(), _p(dm), _p(arg), _p(t_rowptr), _p(t_edge), _p(dbank), ctx.K, cols, ctx.k,
                                             _dt(dm)), "egk_gather_max_bank_grad")
        return df, dbank, None


def gather_max(f, bank, nn):
    """m[n] = max(f[n], bank[nn[n, :]]) elementwise (f32 bank; its rows receive a gradient when the bank is trainable)."""
    return _GatherMax.apply(f, bank, nn)


# ---- per-sequence max pool --------------------------------------------------------------------------------
class _SegMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ptr):
        _need_gpu(x, ptr)
        x = _c(x)
        rows, cols = x.shape
        n_seg = ptr.numel() - 1
        out = torch.empty((n_seg, cols), dtype=x.dtype, device=x.device)
        arg = torch.empty((n_seg, cols), dtype=torch.int32, device=x.device)
        _ck(_lib.load().egk_segment_max_fwd(_stream(), _p(x), _p(ptr), _p(out), _p(arg), n_seg, cols, _dt(x)),
            "egk_segment_max_fwd")
        ctx.rows = rows
        ctx.save_for_backward(arg, ptr)
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, ptr = ctx.saved_tensors
        dout = _c(dout)
        n_seg, cols = dout.shape
        dx = torch.empty((ctx.rows, cols), dtype=dout.dtype, device=dout.device)
        _ck(_lib.load().egk_segment_max_bwd(_stream(), _p(dout), _p(arg), _p(ptr), _p(dx), n_seg, ctx.rows, cols, _dt(dout)),
            "egk_segment_max_bwd")
        return dx, None


def segment_max(x, ptr):
    """global_max_pool over contiguous sequences; ``ptr`` int32 [B+1]."""
    return _SegMax.apply(x, ptr)


class _SegMaxMulti(torch.autograd.Function):
    """``segment_max`` of up to four inputs of one shape over the same sequences: ONE launch forward, ONE backward."""

    @staticmethod
    def forward(ctx, ptr, *xs):
        _need_gpu(ptr, *xs)
        xs = [_c(x) for x in xs]
        rows, cols = xs[0].shape
        n_seg = ptr.numel() - 1
        outs = [torch.empty((n_seg, cols), dtype=xs[0].dtype, device=xs[0].device) for _ in xs]
        args = torch.empty((len(xs), n_seg, cols), dtype=torch.int32, device=xs[0].device)
        _ck(_lib.load().egk_segment_max_multi_fwd(_stream(), _ptr_array(xs), _p(ptr), _ptr_array(outs), _ptr_array(list(args)), len(xs), n_seg,
                                                  cols, _dt(xs[0])), "egk_segment_max_multi_fwd")
        ctx.rows, ctx.n = rows, len(xs)
        ctx.save_for_backward(args, ptr)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        args, ptr = ctx.saved_tensors
        douts = [_c(d) for d in douts]
        n_seg, cols = douts[0].shape
        dxs = [torch.empty((ctx.rows, cols), dtype=douts[0].dtype, device=douts[0].device) for _ in douts]
        _ck(_lib.load().egk_segment_max_multi_bwd(_stream(), _ptr_array(douts), _ptr_array(list(args)), _p(ptr), _ptr_array(dxs), ctx.n, n_seg,
                                                  ctx.rows, cols, _dt(douts[0])), "egk_segment_max_multi_bwd")
        return (None, *dxs)


def segment_max_multi(xs, ptr):
    """[segment_max(x, ptr) for x in xs] -- one launch each way when the inputs share shape, element type and device (2 .. 4 of
    them), the pools one by one otherwise."""
    xs = list(xs)
    x0 = xs[0]
    if (2 <= len(xs) <= 4 and x0.is_cuda and all(x.shape == x0.shape and x.dtype == x0.dtype and x.dim() == 2 for x in xs)
            and switches.enabled("segmax_multi")):
        return list(_SegMaxMulti.apply(ptr, *xs))
    return [segment_max(x, ptr) for x in xs]


# ---- losses ---------------------------------------------------------------------------------------------
def _grad_dtype_of(t: torch.Tensor) -> torch.dtype:
    return getattr(t, "_egk_grad_dtype", torch.float32)


_loss_seed = {"coef": None}


class loss_seed:
    """``with loss_seed(c):`` -- inside, the caller guarantees that the loss vector computed next is back-propagated with the
    constant gradient ``c`` for every element (the engine's training heads: objective = sum_t w_t * mean(loss_t), so
    c = w_t / numel).  A multi-head cross entropy whose logits come from ``classifier_bank`` then computes loss AND gradient
    in one launch (egk_ce_fused) and its backward is a no-op."""

    def __init__(self, coef):
        self.coef = coef

    def __enter__(self):
        self.prev, _loss_seed["coef"] = _loss_seed["coef"], self.coef

    def __exit__(self, *a):
        _loss_seed["coef"] = self.prev


class _CE(torch.autograd.Function):
    @staticmethod
    def _fused(ctx, smoothing, y, logits):
        """One launch for loss and gradient when the seed is known and every head's logits are blocks of ONE bank buffer."""
        if _loss_seed["coef"] is None or not _bank_handoff["on"] or y.dim() != 2 or y.shape[1] != len(logits) or len(logits) > 4:
            return None
        dsts = [getattr(l, "_egk_grad_dst", None) for l in logits]
        if any(d is None for d in dsts) or any(d[0] is not dsts[0][0] or d[2] is not dsts[0][2] for d in dsts):
            return None
        gbuf, state = dsts[0][0], dsts[0][2]
        if any(d[1] in state["filled"] for d in dsts) or any(l.dtype != torch.float32 or l.stride(1) != 1 for l in logits):
            return None
        lib = _lib.load()
        rows, n = logits[0].shape[0], len(logits)
        starts = sorted(d[1] for d in dsts)
        if [d[1] for d in dsts] != starts:
            return None
        ends = starts[1:] + [gbuf.shape[1]]
        loss = torch.empty(rows, dtype=torch.float32, device=gbuf.device)
        lp = (C.c_void_p * n)(*[l.data_ptr() for l in logits])
        ld = (C.c_int64 * n)(*[l.stride(0) for l in logits])
        Cs = (C.c_int32 * n)(*[l.shape[1] for l in logits])
        pad = (C.c_int32 * n)(*[e - s0 for s0, e in zip(starts, ends)])
        dcol = (C.c_int64 * n)(*starts)
        _ck(lib.egk_ce_fused(_stream(), lp, ld, Cs, pad, dcol, n, _p(y), y.shape[1], _p(loss), _p(gbuf), gbuf.stride(0), rows,
                             smoothing, float(_loss_seed["coef"]), _dt(gbuf)), "egk_ce_fused")
        if starts[0] > 0:
            gbuf[:, :starts[0]].zero_()
        state["filled"].update(starts)
        state["pads"] = True
        ctx.fused, ctx.shapes, ctx.seed = True, [tuple(l.shape) for l in logits], float(_loss_seed["coef"])
        return loss

    @staticmethod
    def forward(ctx, smoothing, y, gdt, *logits):
        # loss[n] = sum_h CE(logits[h][n], y[n, h]) with ignore_index -1 (y: [N] or [N, heads] int64)
        _need_gpu(y, *logits)
        lib = _lib.load()
        rows = logits[0].shape[0]
        y = y.contiguous()
        ctx.fused = False
        fused = _CE._fused(ctx, smoothing, y, logits)
        if fused is not None:
            return fused
        loss = torch.empty(rows, dtype=torch.float32, device=logits[0].device)
        ystride = 1 if y.dim() == 1 else y.shape[1]
        saved = []
        ctx.dst = [getattr(l, "_egk_grad_dst", None) if _bank_handoff["on"] else None for l in logits]
        for h, l in enumerate(logits):
            if l.dtype != torch.float32:
                raise TypeError("cross_entropy expects f32 logits")
            l = _rm(l)
            lse = torch.empty(rows, dtype=torch.float32, device=l.device)
            yh = y if y.dim() == 1 else y[:, h]
            _ck(lib.egk_ce_fwd(_stream(), _p(l), l.stride(0), C.c_void_p(yh.data_ptr()), ystride, _p(loss), _p(lse), rows,
                               l.shape[1], smoothing, int(h > 0)), "egk_ce_fwd")
            saved += [l, lse]
        ctx.smoothing, ctx.ystride, ctx.nh, ctx.gdt = smoothing, ystride, len(logits), gdt
        ctx.save_for_backward(y, *saved)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        lib = _lib.load()
        if ctx.fused:  # the gradient is already in the bank's operand buffer: placeholders for autograd
            return (None, None, None, *[torch.empty(s_, dtype=torch.float32, device=gloss.device) for s_ in ctx.shapes])
        y, *saved = ctx.saved_tensors
        gloss = _f32c(gloss)
        grads = []
        for h in range(ctx.nh):
            l, lse = saved[2 * h], saved[2 * h + 1]
            rows, Cn = l.shape
            d = torch.empty((rows, Cn), dtype=torch.float32, device=l.device)
            yh = y if y.dim() == 1 else y[:, h]
            dst = ctx.dst[h]
            if dst is not None and dst[1] not in dst[2]["filled"]:
                # classifier_bank's operand buffer takes the gradient directly; ``d`` stays an uninitialised placeholder
                gbuf, c0, state = dst
                out = gbuf[:, c0:c0 + Cn]
                _ck(lib.egk_ce_bwd(_stream(), _p(l), l.stride(0), C.c_void_p(yh.data_ptr()), ctx.ystride, _p(lse), _p(gloss),
                                   _p(out), gbuf.stride(0), rows, Cn, ctx.smoothing, _dt(gbuf)), "egk_ce_bwd")
                state["filled"].add(c0)
            else:
                _ck(lib.egk_ce_bwd(_stream(), _p(l), l.stride(0), C.c_void_p(yh.data_ptr()), ctx.ystride, _p(lse), _p(gloss),
                                   _p(d), d.stride(0), rows, Cn, ctx.smoothing, _dt(d)), "egk_ce_bwd")
            grads.append(d)
        return (None, None, None, *grads)


def _ce_fused_plan(y, logits):
    """What the fused cross entropy needs of one task -- (gradient operand, its state, column starts, block widths) -- or None
    when it does not apply (see ``_CE._fused``)."""
    if _loss_seed["coef"] is None and not _bank_handoff["on"]:
        return None
    if not _bank_handoff["on"] or y.dim() != 2 or y.shape[1] != len(logits) or len(logits) > 4:
        return None
    dsts = [getattr(l, "_egk_grad_dst", None) for l in logits]
    if any(d is None for d in dsts) or any(d[0] is not dsts[0][0] or d[2] is not dsts[0][2] for d in dsts):
        return None
    gbuf, state = dsts[0][0], dsts[0][2]
    if any(d[1] in state["filled"] for d in dsts) or any(l.dtype != torch.float32 or l.stride(1) != 1 for l in logits):
        return None
    starts = sorted(d[1] for d in dsts)
    if [d[1] for d in dsts] != starts or starts[0] != 0:
        return None
    ends = starts[1:] + [gbuf.shape[1]]
    return gbuf, state, starts, [e - s0 for s0, e in zip(starts, ends)]


class _CEMulti(torch.autograd.Function):
    """The fused cross entropies of several tasks (each: loss + gradient written into its bank's operand buffer, ``_CE._fused``)
    as ONE launch, ``egk_ce_fused_multi``."""

    @staticmethod
    def forward(ctx, smoothing, coefs, heads_per_task, plans, *tensors):
        lib = _lib.load()
        n = len(coefs)
        arr = (_lib.CETask * n)()
        losses, shapes, k = [], [], 0
        for i in range(n):
            y = tensors[k]
            logits = tensors[k + 1:k + 1 + heads_per_task[i]]
            k += 1 + heads_per_task[i]
            gbuf, state, starts, pads = plans[i]
            rows = logits[0].shape[0]
            loss = torch.empty(rows, dtype=torch.float32, device=gbuf.device)
            t = arr[i]
            for h, l in enumerate(logits):
                t.logits[h], t.ld[h], t.C[h], t.pad[h], t.dcol[h] = l.data_ptr(), l.stride(0), l.shape[1], pads[h], starts[h]
            t.n_heads, t.y, t.y_stride, t.loss = len(logits), y.data_ptr(), y.shape[1], loss.data_ptr()
            t.dlogits, t.ldd, t.rows, t.gscale = gbuf.data_ptr(), gbuf.stride(0), rows, float(coefs[i])
            losses.append(loss)
            shapes.append([tuple(l.shape) for l in logits])
            state["filled"].update(starts)
            state["pads"] = True
        _ck(lib.egk_ce_fused_multi(_stream(), arr, n, float(smoothing), _dt(plans[0][0])), "egk_ce_fused_multi")
        ctx.shapes, ctx.heads = shapes, heads_per_task
        ctx.set_materialize_grads(False)
        return tuple(losses)

    @staticmethod
    def backward(ctx, *glosses):
        out = [None, None, None, None]
        dev = next(g.device for g in glosses if g is not None)
        for shp in ctx.shapes:  # placeholders: the gradients are already in the banks' operand buffers
            out.append(None)  # y
            out.extend(torch.empty(s_, dtype=torch.float32, device=dev) for s_ in shp)
        return tuple(out)


def cross_entropy_multi(tasks, coefs, smoothing: float = 0.0):
    """[loss_t] of ``cross_entropy(logits_t, y_t, smoothing)`` for several tasks whose backward seeds ``coefs`` are known, in ONE
    launch that also writes every task's gradient operand -- or None when some task does not qualify for the fused form
    (the caller then takes them one by one).  ``tasks``: [(logits tuple, y [N, heads] int64)]."""
    if not (2 <= len(tasks) <= 4) or not torch.is_grad_enabled():
        return None
    plans, flat, heads = [], [], []
    dt = None
    for logits, y in tasks:
        if torch.is_tensor(logits):
            logits = (logits,)
        y = y.contiguous()
        plan = _ce_fused_plan(y, logits)
        if plan is None or (dt is not None and plan[0].dtype != dt):
            return None
        dt = plan[0].dtype
        plans.append(plan)
        heads.append(len(logits))
        flat += [y, *logits]
    return list(_CEMulti.apply(float(smoothing), [float(c) for c in coefs], heads, plans, *flat))


def cross_entropy(logits, y, smoothing: float = 0.0):
    """Per-row CrossEntropy(reduction='none', ignore_index=-1), summed over heads when ``logits`` is
    a tuple and y is [N, heads].  Logits are f32; their gradient is emitted in the element type the
    producing contraction wants (bf16 in 'bf16' mode)."""
    if torch.is_tensor(logits):
        logits = (logits,)
    return _CE.apply(float(smoothing), y, _state["act"], *logits)


class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, gdt):
        _need_gpu(logits, y)
        logits = _f32c(logits)
        y = y.contiguous()
        loss = torch.empty_like(logits)
        _ck(_lib.load().egk_bce_fwd(_stream(), _p(logits), _p(y), _p(loss), logits.numel()), "egk_bce_fwd")
        ctx.gdt = gdt
        ctx.save_for_backward(logits, y)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, y = ctx.saved_tensors
        g = _f32c(g)
        d = torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
        _ck(_lib.load().egk_bce_bwd(_stream(), _p(logits), _p(y), _p(g), _p(d), logits.numel(), _dt(d)), "egk_bce_bwd")
        return d, None, None


def bce_with_logits(logits, y):
    """BCEWithLogitsLoss(reduction='none') against y.float(); y int64 of the same shape."""
    if y.dtype != torch.int64:
        y = y.to(torch.int64)
    return _BCE.apply(logits, y, _state["act"])


class _RowDotBCE(torch.autograd.Function):
    """Linear(features, 1) + squeeze + BCE-with-logits(reduction='none') as ONE row pass that also emits the gradients
    (egk_rowdot_bce): used when the seed of the loss vector's backward is known (``loss_seed``)."""

    @staticmethod
    def forward(ctx, f, W, b, y, seed):
        _need_gpu(f, W, y)
        lib = _lib.load()
        f = _c(f)
        rows, cols = f.shape
        w_op = weight_operand(W, f.dtype).reshape(-1)
        bias = _f32c(b) if b is not None else None
        y = y.contiguous()
        logits = torch.empty(rows, dtype=torch.float32, device=f.device)
        loss = torch.empty(rows, dtype=torch.float32, device=f.device)
        df = torch.empty_like(f)
        ws = torch.empty(lib.egk_rowdot_ws_rows(rows) * (cols + 4), dtype=torch.float32, device=f.device)
        _ck(lib.egk_rowdot_bce(_stream(), _p(f), _p(w_op), _p(bias), _p(y), _p(logits), _p(loss), _p(df), _p(ws), rows, cols,
                               float(seed), _dt(f)), "egk_rowdot_bce")
        slot_w, slot_b = _grad_slot(W), _grad_slot(b)
        dw = slot_w if slot_w is not None else torch.zeros(W.shape, dtype=torch.float32, device=f.device)
        db = (slot_b if slot_b is not None else torch.zeros(b.shape, dtype=torch.float32, device=f.device)) if b is not None else None
        _ck(lib.egk_rowdot_reduce(_stream(), _p(ws), _p(dw), _p(db), rows, cols), "egk_rowdot_reduce")
        ctx.ret = (df, None if slot_w is not None else dw, None if (slot_b is not None or b is None) else db)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)  # (no zero tensor for the logits' absent gradient: a launch of its own)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        df, dw, db = ctx.ret  # computed in forward from the announced seed (the constant weight / numel of the objective)
        ctx.ret = None        # (sole owner of df from here: autograd keeps it as the leaf's gradient instead of cloning it)
        return df, dw, db, None, None


def linear1_bce_ok(f, W) -> bool:
    """The one-pass head applies: the backward seed is announced (``loss_seed``), gradients are wanted, ``f`` is a device
    matrix of at most 4096 columns in an element type the row kernels compute in (bf16 activations, or exact-f32 mode)."""
    return bool(_loss_seed["coef"] is not None and torch.is_grad_enabled() and f.is_cuda and f.dim() == 2 and W.dim() == 2
                and W.shape[0] == 1 and W.shape[1] == f.shape[1] and f.shape[1] <= 4096 and f.shape[1] % 8 == 0
                and (f.dtype == torch.bfloat16 or (f.dtype == torch.float32 and _state["compute"] == F32)))


def linear1_bce(f, W, b, y):
    """(loss [N], logits [N]) of BCEWithLogits(reduction='none')(Linear(H, 1)(f).squeeze(), y.float()) -- reference
    models/tasks/pnr.py:37-52 + main_temporal.py:117-121 -- in one row pass that also writes d f, d W, d b.  Requires
    ``linear1_bce_ok``."""
    if not linear1_bce_ok(f, W):
        raise RuntimeError("linear1_bce: needs an announced loss seed (ops.loss_seed) and a device feature matrix")
    if y.dtype != torch.int64:
        y = y.to(torch.int64)
    return _RowDotBCE.apply(f, W, b, y, float(_loss_seed["coef"]))


class _RowDotCE2(torch.autograd.Function):
    """Linear(features, 2) + cross entropy (reduction 'none', ignore_index -1) over FEW rows as ONE launch that also emits the
    gradients (egk_rowdot_ce2): used when the seed of the loss vector's backward is known (``loss_seed``)."""

    @staticmethod
    def forward(ctx, f, W, b, y, smoothing, seed):
        _need_gpu(f, W, y)
        lib = _lib.load()
        f = _c(f)
        rows, cols = f.shape
        w_op = _c(weight_operand(W, f.dtype))
        bias = _f32c(b) if b is not None else None
        y = y.contiguous()
        logits = torch.empty(rows, 2, dtype=torch.float32, device=f.device)
        loss = torch.empty(rows, dtype=torch.float32, device=f.device)
        df = torch.empty_like(f)
        slot_w, slot_b = _grad_slot(W), _grad_slot(b)
        dw = slot_w if slot_w is not None else torch.zeros(W.shape, dtype=torch.float32, device=f.device)
        db = (slot_b if slot_b is not None else torch.zeros(b.shape, dtype=torch.float32, device=f.device)) if b is not None else None
        gws = torch.empty(rows, 2, dtype=torch.float32, device=f.device)
        _ck(lib.egk_rowdot_ce2(_stream(), _p(f), _p(w_op), _p(bias), _p(y), _p(logits), _p(loss), _p(df), _p(dw), _p(db), _p(gws), rows,
                               cols, float(smoothing), float(seed), _dt(f)), "egk_rowdot_ce2")
        ctx.ret = (df, None if slot_w is not None else dw, None if (slot_b is not None or b is None) else db)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        df, dw, db = ctx.ret  # computed in forward from the announced seed (the constant weight / numel of the objective)
        ctx.ret = None
        return df, dw, db, None, None, None


class _RowDotCE2Multi(torch.autograd.Function):
    """``_RowDotCE2`` over n sources with their own classifiers: logits = (mean | sum)_k Linear_k(f_k) (egk_rowdot_ce2_multi)."""

    @staticmethod
    def forward(ctx, y, smoothing, average, seed, n, *tensors):
        fs, Ws, bs = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        _need_gpu(fs[0], y)
        lib = _lib.load()
        fs = [_c(f) for f in fs]
        rows, cols = fs[0].shape
        w_ops = [_c(weight_operand(W, fs[0].dtype)) for W in Ws]
        biases = [(_f32c(b) if b is not None else None) for b in bs]
        y = y.contiguous()
        dev = fs[0].device
        logits = torch.empty(rows, 2, dtype=torch.float32, device=dev)
        loss = torch.empty(rows, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[5:]
        need_f, need_w, need_b = need[:n], need[n:2 * n], need[2 * n:3 * n]
        dfs = [torch.empty_like(f) if nf else None for f, nf in zip(fs, need_f)]
        slots_w, slots_b = [_grad_slot(W) for W in Ws], [_grad_slot(b) for b in bs]
        dws = [(sw if sw is not None else torch.zeros(W.shape, dtype=torch.float32, device=dev)) if nw else None
               for W, sw, nw in zip(Ws, slots_w, need_w)]
        dbs = [((sb if sb is not None else torch.zeros(b.shape, dtype=torch.float32, device=dev)) if (b is not None and nb) else None)
               for b, sb, nb in zip(bs, slots_b, need_b)]
        arr = lambda ts: (C.c_void_p * n)(*[(t.data_ptr() if t is not None else 0) for t in ts])
        gws = torch.empty(rows, 2, dtype=torch.float32, device=dev)

        def launch(phase):  # (0: both launches; 1: the row launch -- loss, logits, d f; 2: the column launch -- d W, d b)
            _ck(lib.egk_rowdot_ce2_multi(_stream(), n, arr(fs), arr(w_ops), arr(biases), _p(y), _p(logits), _p(loss), arr(dfs), arr(dws),
                                         arr(dbs), _p(gws), rows, cols, int(bool(average)) | (phase << 1), float(smoothing), float(seed),
                                         _dt(fs[0])), "egk_rowdot_ce2_multi")
        in_slots = all((dw is None or sw is not None) for dw, sw in zip(dws, slots_w)) and all(
            (db is None or sb is not None) for db, sb in zip(dbs, slots_b))
        if in_slots and any(d is not None for d in (*dws, *dbs)) and switches.enabled("ce2_cols_ride"):
            # the classifiers' gradients feed nothing on the chain and land in the optimizer's slots: their launch (27 us on the
            # critical chain of BASELINE config 4 between the head and backward) rides with the next flush of the parked weight
            # gradients (``park_rider``: at once when nothing can be parked)
            launch(1)
            park_rider(lambda: launch(2), hold=(*fs, gws, *w_ops))
        else:
            launch(0)
        ctx.keep = (fs, w_ops, biases, gws)
        ctx.ret = (dfs, [None if (sw is not None) else dw for dw, sw in zip(dws, slots_w)],
                   [None if (sb is not None) else db for db, sb in zip(dbs, slots_b)])
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        dfs, dws, dbs = ctx.ret  # computed in forward from the announced seed
        ctx.ret = ctx.keep = None
        return (None, None, None, None, None, *dfs, *dws, *dbs)


def linear2_ce_multi(fs, Ws, bs, y, smoothing: float = 0.0, average: bool = False):
    """(loss [R], logits [R, 2]) of CrossEntropy(reduction='none', ignore_index=-1, label_smoothing)(fuse_k Linear_k(f_k), y) with
    fuse = mean (``average``) or sum over the sources -- reference models/tasks/oscc.py:65-79 with ``aux_features`` -- in one launch
    that also writes every d f_k, d W_k, d b_k.  Requires ``linear2_ce_ok`` for every source."""
    if not all(linear2_ce_ok(f.shape[0], f, W) for f, W in zip(fs, Ws)) or not (1 <= len(fs) <= 4):
        raise RuntimeError("linear2_ce_multi: needs an announced loss seed (ops.loss_seed) and 1 .. 4 small device feature matrices")
    if y.dtype != torch.int64:
        y = y.to(torch.int64)
    return _RowDotCE2Multi.apply(y, float(smoothing), bool(average), float(_loss_seed["coef"]), len(fs), *fs, *Ws, *bs)


def linear2_ce_ok(rows: int, f, W) -> bool:
    """The one-launch two-logit head applies to ``rows`` pooled rows of the width / element type of ``f``: the backward seed is
    announced (``loss_seed``), gradients are wanted, few rows, a width the row kernels take."""
    return bool(_loss_seed["coef"] is not None and torch.is_grad_enabled() and f.is_cuda and f.dim() == 2 and W.dim() == 2
                and W.shape[0] == 2 and W.shape[1] == f.shape[1] and f.shape[1] <= 4096 and f.shape[1] % 8 == 0
                and 0 < rows <= _lib.load().egk_rowdot_ce2_max_rows()
                and (f.dtype == torch.bfloat16 or (f.dtype == torch.float32 and _state["compute"] == F32)))


def linear2_ce(f, W, b, y, smoothing: float = 0.0):
    """(loss [R], logits [R, 2]) of CrossEntropy(reduction='none', ignore_index=-1, label_smoothing)(Linear(H, 2)(f), y) --
    reference models/tasks/oscc.py:65-79 + main_temporal.py:291 -- in one launch that also writes d f, d W, d b.  Requires
    ``linear2_ce_ok``."""
    if not linear2_ce_ok(f.shape[0], f, W):
        raise RuntimeError("linear2_ce: needs an announced loss seed (ops.loss_seed) and a small device feature matrix")
    if y.dtype != torch.int64:
        y = y.to(torch.int64)
    return _RowDotCE2.apply(f, W, b, y, float(smoothing), float(_loss_seed["coef"]))


class _OneHotSigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, kind, alpha, gamma):
        _need_gpu(logits, y)
        logits = _f32c(logits)
        y = y.contiguous()
        rows, Cn = logits.shape
        loss = torch.empty_like(logits)
        _ck(_lib.load().egk_onehot_sigmoid_loss_fwd(_stream(), _p(logits), _p(y), _p(loss), rows, Cn, kind, alpha, gamma),
            "egk_onehot_sigmoid_loss_fwd")
        ctx.cfg = (kind, alpha, gamma)
        ctx.save_for_backward(logits, y)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, y = ctx.saved_tensors
        g = _f32c(g)
        rows, Cn = logits.shape
        d = torch.empty_like(logits)
        kind, alpha, gamma = ctx.cfg
        _ck(_lib.load().egk_onehot_sigmoid_loss_bwd(_stream(), _p(logits), _p(y), _p(g), _p(d), rows, Cn, kind, alpha, gamma,
                                                    _dt(d)), "egk_onehot_sigmoid_loss_bwd")
        return d, None, None, None, None


def onehot_bce_with_logits(logits, y):
    """F.binary_cross_entropy_with_logits(logits, one_hot(y, C).float(), reduction='none') (reference oscc.py:91-93)."""
    if y.dtype != torch.int64:
        y = y.to(torch.int64)
    return _OneHotSigmoid.apply(logits, y, 0, -1.0, 0.0)


def onehot_sigmoid_focal_loss(logits, y, alpha: float = 0.5, gamma: float = 2.0):
    """torchvision.ops.sigmoid_focal_loss(logits, one_hot(y, C).float(), alpha, gamma, reduction='none')
    (reference oscc.py:94-96)."""
    if y.dtype != torch.int64:
        y = y.to(torch.int64)
    return _OneHotSigmoid.apply(logits, y, 1, float(alpha), float(gamma))


# ---- dropout / reductions -----------------------------------------------------------------------------------
class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        _need_gpu(x)
        x = _c(x)
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        seed, off = _next_rng(x.numel())
        _ck(_lib.load().egk_dropout_fwd(_stream(), _p(x), _p(y), _p(mask), x.numel(), p, seed, off,
                                        _p(rng_device_offset(x.device)), _dt(x)), "egk_dropout_fwd")
        ctx.p = p
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _ck(_lib.load().egk_dropout_bwd(_stream(), _p(dy), _p(mask), _p(dx), dy.numel(), ctx.p, _dt(dy)), "egk_dropout_bwd")
        return dx, None


def dropout(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, float(p))


class _Relu(torch.autograd.Function):
    """max(x, 0) as a launch of its own (where no contraction epilogue can take it: the input ReLU of the multi-scale relation
    module, reference models/TRN.py:32-36).  Forward and backward are the same gate kernel: y = x > 0 ? x : 0, dx = y > 0 ? dy : 0."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = _c(x)
        y = torch.empty_like(x)
        _ck(_lib.load().egk_relu_gate(_stream(), _p(x), _p(x), _p(y), x.numel(), _dt(x)), "egk_relu_gate")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _match(_c(dy), y.dtype)
        dx = torch.empty_like(y)
        _ck(_lib.load().egk_relu_gate(_stream(), _p(dy), _p(y), _p(dx), y.numel(), _dt(y)), "egk_relu_gate")
        return dx


def relu(x):
    return _Relu.apply(x)


class _WeightedMeanSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, counts, *vectors):
        _need_gpu(*vectors)
        import ctypes as C_
        lib = _lib.load()
        k = len(vectors)
        vs = [_f32c(v) for v in vectors]
        coefs = [w / max(v.numel() if c is None else c, 1) for w, v, c in zip(weights, vs, counts)]
        out = torch.empty(1, dtype=torch.float32, device=vs[0].device)
        xs = (C_.c_void_p * k)(*[v.data_ptr() if v.numel() else None for v in vs])
        ns = (C_.c_int64 * k)(*[v.numel() for v in vs])
        cf = (C_.c_float * k)(*coefs)
        _ck(lib.egk_weighted_sums(_stream(), xs, ns, cf, k, _p(out)), "egk_weighted_sums")
        ctx.coefs, ctx.shapes = coefs, [v.shape for v in vectors]
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        import ctypes as C_
        lib = _lib.load()
        g = _f32c(g.reshape(1))
        k = len(ctx.shapes)
        grads = [torch.empty(shape, dtype=torch.float32, device=g.device) for shape in ctx.shapes]
        outs = (C_.c_void_p * k)(*[d.data_ptr() if d.numel() else None for d in grads])
        ns = (C_.c_int64 * k)(*[d.numel() for d in grads])
        cf = (C_.c_float * k)(*ctx.coefs)
        _ck(lib.egk_fill_scaled_multi(_stream(), _p(g), cf, outs, ns, k), "egk_fill_scaled_multi")
        return (None, None, *grads)


@torch.no_grad()
def weighted_mean_sum_into(out, vectors, weights, counts=None, acc=None):
    """``weighted_mean_sum`` without autograd, written into ``out`` (an f32 tensor of one element).  ``acc`` (f64 [len(vectors)],
    optional): acc[k] += vectors[k].sum() in the same launch -- the running loss sums of a training loop (a ``None`` vector adds 0)."""
    import ctypes as C_
    k = len(vectors)
    vs = [_f32c(v) if v is not None else None for v in vectors]
    counts = counts if counts is not None else [None] * k
    coefs = [0.0 if v is None else float(w) / max(v.numel() if c is None else c, 1) for w, v, c in zip(weights, vs, counts)]
    xs = (C_.c_void_p * k)(*[v.data_ptr() if (v is not None and v.numel()) else None for v in vs])
    ns = (C_.c_int64 * k)(*[v.numel() if v is not None else 0 for v in vs])
    cf = (C_.c_float * k)(*coefs)
    if acc is None:
        _ck(_lib.load().egk_weighted_sums(_stream(), xs, ns, cf, k, _p(out)), "egk_weighted_sums")
    else:
        if acc.dtype != torch.float64 or acc.numel() < k or not acc.is_contiguous():
            raise ValueError("weighted_mean_sum_into: acc must be a contiguous float64 tensor with one element per vector")
        _ck(_lib.load().egk_weighted_sums_acc(_stream(), xs, ns, cf, k, _p(out), _p(acc)), "egk_weighted_sums_acc")
    return out


def weighted_mean_sum(vectors, weights, counts=None):
    """sum_i weights[i] * vectors[i].mean()  -- the training objective of main_temporal.py:99-128
    (``torch.stack([w * l.mean() ...]).sum()``) as deterministic single-workgroup reductions.  ``counts[i]`` (optional): the
    number of elements the mean of vector i divides by, when the vector holds only the non-zero ones (a compacted head's loss
    vector: the ignored nodes' zeros are left out of the sum, not of the mean)."""
    counts = tuple(counts) if counts is not None else (None,) * len(vectors)
    return _WeightedMeanSum.apply(tuple(float(w) for w in weights), counts, *vectors)


class _SumTensors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, *ts):
        _need_gpu(*ts)
        lib = _lib.load()
        ts = [_f32c(t) for t in ts]
        out = torch.empty_like(ts[0])
        n = out.numel()
        if len(ts) == 1:
            _ck(lib.egk_axpby(_stream(), _p(ts[0]), None, _p(out), n, scale, 0.0), "egk_axpby")
        else:
            _ck(lib.egk_axpby(_stream(), _p(ts[0]), _p(ts[1]), _p(out), n, scale, scale), "egk_axpby")
            for t in ts[2:]:
                _ck(lib.egk_axpby(_stream(), _p(out), _p(t), _p(out), n, 1.0, scale), "egk_axpby")
        ctx.scale, ctx.n = scale, len(ts)
        gd = [_grad_dtype_of(t) for t in ts]
        if all(d == gd[0] for d in gd):
            out._egk_grad_dtype = gd[0]
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.scale == 1.0:
            return (None, *([g] * ctx.n))
        g = _c(g)
        d = cast_raw(g, torch.float32) if g.dtype != torch.float32 else g
        o = torch.empty_like(d)
        _ck(_lib.load().egk_axpby(_stream(), _p(d), None, _p(o), d.numel(), ctx.scale, 0.0), "egk_axpby")
        return (None, *([o] * ctx.n))


def sum_tensors(ts, scale: float = 1.0):
    """scale * sum(ts): logit fusion ``stack([...]).sum(0)`` / ``.mean(0)`` without the stacked copy (f32 logits)."""
    return _SumTensors.apply(float(scale), *ts)


class _SplitRows(torch.autograd.Function):
    """Row slices of the fused backbone output, one per task batch.  Backward writes the incoming slice
    gradients into ONE buffer (a copy per slice) instead of autograd's fill + copy + add per slice."""

    @staticmethod
    def forward(ctx, x, sizes):
        ctx.sizes, ctx.shape, ctx.dtype = sizes, x.shape, x.dtype
        outs, off = [], 0
        for s in sizes:
            outs.append(x[off:off + s])
            off += s
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        import ctypes as C_
        lib = _lib.load()
        dev = next(g.device for g in gs if g is not None)
        # the slice gradients already are consecutive row ranges of ONE buffer (the grouped projection heads write their
        # dx that way): hand that buffer back, no copy
        if all(g is not None and g.dtype == ctx.dtype and g.is_contiguous() for g in gs):
            st = gs[0].untyped_storage()
            row_elems = 1
            for d_ in ctx.shape[1:]:
                row_elems *= d_
            off, ok = gs[0].storage_offset(), True
            for g, sz in zip(gs, ctx.sizes):
                ok = (ok and g.untyped_storage().data_ptr() == st.data_ptr() and g.storage_offset() == off
                      and tuple(g.shape) == (sz, *ctx.shape[1:]))
                off += sz * row_elems
            if ok and (off - gs[0].storage_offset()) == ctx.shape[0] * row_elems:
                whole = torch.empty(0, dtype=ctx.dtype, device=dev).set_(st, gs[0].storage_offset(), tuple(ctx.shape))
                return whole, None
        out = torch.empty(ctx.shape, dtype=ctx.dtype, device=dev)
        row_bytes = out[0].numel() * out.element_size() if out.shape[0] else 0
        srcs = [None if g is None else _c(_match(g, ctx.dtype)) for g in gs]
        k = len(srcs)
        if k <= 8:  # one launch for all slices (a copy per slice was a chain of memcpy nodes)
            ptrs = (C_.c_void_p * k)(*[s_.data_ptr() if (s_ is not None and s_.numel()) else None for s_ in srcs])
            nb = (C_.c_int64 * k)(*[s * row_bytes for s in ctx.sizes])
            _ck(lib.egk_copy_blocks(_stream(), ptrs, nb, _p(out), k), "egk_copy_blocks")
            return out, None
        off = 0
        for s, g in zip(ctx.sizes, srcs):
            dst = out[off:off + s]
            dst.zero_() if g is None else dst.copy_(g)
            off += s
        return out, None


def split_rows(x, sizes):
    return _SplitRows.apply(x, tuple(int(s) for s in sizes))


# ---- cosine k-NN (no grad) ------------------------------------------------------------------------------------
def to_act_f32(x):
    """f32 copy of a bf16 activation that remembers its source (a three-product contraction then skips the zero low half)."""
    y = cast_raw(x, torch.float32)
    if x.dtype == torch.bfloat16:
        y._egk_bf16_src = x if x.is_contiguous() else x.contiguous()
    return y


@torch.no_grad()
def row_inv_norm(x):
    _need_gpu(x)
    x = _c(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _ck(_lib.load().egk_row_inv_norm(_stream(), _p(x), _p(out), x.shape[0], x.shape[1], _dt(x)), "egk_row_inv_norm")
    return out


@torch.no_grad()
def scaled_one_minus(dot, f_inv, b_inv):
    _need_gpu(dot)
    out = torch.empty_like(dot)
    _ck(_lib.load().egk_cos_dist(_stream(), _p(dot), dot.stride(0), _p(f_inv), _p(b_inv), _p(out), dot.shape[0],
                                 dot.shape[1]), "egk_cos_dist")
    return out


@torch.no_grad()
def row_sq_norm(x):
    _need_gpu(x)
    x = _c(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _ck(_lib.load().egk_row_sq_norm(_stream(), _p(x), _p(out), x.shape[0], x.shape[1], _dt(x)), "egk_row_sq_norm")
    return out


# The one-product search (egk_topk_window): per bank the bf16 operand hi(P) and the largest rounding residual ratio of its rows,
# kept while (address, shape, version) stand -- the banks are frozen (graphONE.py:48) unless GraphONE is built with freeze=False.
_window_search = {"on": switches.enabled("window_search")}
_window_f16 = {"on": switches.enabled("window_f16")}  # the grouped search's screen on the f16 matrix instructions
_window_bank_cache = {}
_window_stats = {"cand": None}  # development / tests: an int32 [N] tensor here receives the candidates per row of the next search


def _window_search_ok(N, K, H, k, f, bank) -> bool:
    return bool(_window_search["on"] and H % 64 == 0 and H >= 64 and K >= 64 and k <= 16 and f.stride(0) % 4 == 0
                and bank.stride(0) % 4 == 0 and f.data_ptr() % 16 == 0 and bank.data_ptr() % 16 == 0 and N > 0)


def _cast_f16_bits(x: torch.Tensor) -> torch.Tensor:
    """IEEE-half rounding of an f32 matrix, held in a bf16-TYPED tensor (16-bit words: the contraction descriptors know one 16-bit
    storage type; ``op_f16`` tells the launch what the words mean)."""
    x = _c(x)
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _ck(_lib.load().egk_cast_f16(_stream(), _p(x), _p(y), x.numel()), "egk_cast_f16")
    return y


def _bank_window_operand(bank, f16: bool = False):
    key = (bank.data_ptr(), tuple(bank.shape), bank.device.index, bool(f16))
    hit = _window_bank_cache.get(key)
    if hit is not None and hit[0] == bank._version:
        return hit[1], hit[2]
    if len(_window_bank_cache) > 64:
        _window_bank_cache.clear()
    hi = _cast_f16_bits(bank) if f16 else cast_raw(bank, torch.bfloat16)
    r = torch.empty(bank.shape[0], dtype=torch.float32, device=bank.device)
    rmax = torch.empty(1, dtype=torch.float32, device=bank.device)
    _ck(_lib.load().egk_residual_ratio16(_stream(), _p(bank), bank.stride(0), _p(r), _p(rmax), bank.shape[0], bank.shape[1], int(f16)),
        "egk_residual_ratio16")
    if not torch.cuda.is_current_stream_capturing():
        # (tensors made inside a hipGraph capture live in the graph's pool and hold values only after a replay: never cached --
        #  as ``weight_operand`` does with the frozen weights' copies)
        _window_bank_cache[key] = (bank._version, hi, rmax, bank)  # (the bank itself: its address cannot be reused while it is cached)
    return hi, rmax


@torch.no_grad()
def nearest_prototypes(f, bank, k, distance_func: str = "cosine", bank_norm=None):
    """Indices [N, k] (int64, ascending distance, ties to the lower index) of the k nearest bank rows of every row of
    f (GraphONE.__compute_edges, reference graphONE.py:119-141).  'cosine': 1 - cos similarity; 'l2': cdist / 4096.
    The f . bank^T product always runs on the exact-f32 MFMA path (bf16 features are widened first) so that index
    selection depends on the stored feature values only, not on the MFMA type.  ``bank_norm``: cached per-row
    1/||p|| (cosine) or ||p||^2 (l2) of the bank."""
    _need_gpu(f, bank)
    if distance_func not in ("cosine", "l2"):
        raise ValueError(f"Unknown distance function: {distance_func}")
    lib = _lib.load()
    f, bank = _c(f), _f32c(bank)
    if f.dtype != torch.float32:
        f = to_act_f32(f)
    N, H = f.shape
    K = bank.shape[0]
    l2 = distance_func == "l2"
    norm = row_sq_norm if l2 else row_inv_norm
    if bank_norm is None:
        bank_norm = norm(bank)
    f_norm = norm(f)
    dot = torch.empty((N, K), dtype=torch.float32, device=f.device)
    nn = torch.empty((N, k), dtype=torch.int64, device=f.device)
    if not l2 and _state["compute"] != F32 and _window_search_ok(N, K, H, k, f, bank):
        # the bf16 modes: ONE bf16 product of the rounded operands, a proven error window per row, the exact distances of the few
        # prototypes inside it (egk_topk_window) -- a third of the three-product search's matrix work, the same lists
        hi = cast_raw(f, torch.bfloat16)
        bank_hi, rb_max = _bank_window_operand(bank)
        gemm(N, K, hi, H, bank_hi, H, H, dot, K, compute=BF16)
        cand = _window_stats["cand"]
        if cand is not None and (cand.numel() != N or cand.device != f.device):
            cand = None
        _ck(lib.egk_topk_window(_stream(), _p(dot), K, _p(f), f.stride(0), _p(bank), bank.stride(0), _p(f_norm), _p(bank_norm),
                                _p(rb_max), _p(nn), _p(cand), N, K, H, k), "egk_topk_window")
        return nn
    # 'f32' mode: the exact-f32 matrix instructions.  The bf16 modes: the three-product contraction of the f32 values
    # (absolute error of a cosine ~3e-7, against ranking gaps of ~5e-3 between neighbouring prototypes: the lists are those of
    # the exact product wherever its gap exceeds 1e-5) at a fifth of the time -- the N x K x H product was 1 ms of the 3.4 ms
    # EgoPack step on the exact path
    gemm(N, K, f, H, bank, H, H, dot, K, compute=F32 if _state["compute"] == F32 else X3)
    fn = lib.egk_topk_smallest_l2 if l2 else lib.egk_topk_smallest
    _ck(fn(_stream(), _p(dot), K, _p(f_norm), _p(bank_norm), _p(nn), N, K, k), "egk_topk_smallest")
    return nn


def rows_of_one_buffer(feats):
    """The [sum rows, H] buffer whose consecutive row blocks ``feats`` are (contiguous, in order, nothing else in it), or None."""
    f0 = feats[0]
    base = f0._base
    if base is None or base.dim() != 2 or not base.is_contiguous() or base.shape != (sum(f.shape[0] for f in feats), f0.shape[1]):
        return None
    off = base.data_ptr()
    for f in feats:
        if f._base is not base or f.dim() != 2 or not f.is_contiguous() or f.data_ptr() != off:
            return None
        off += f.shape[0] * f.shape[1] * base.element_size()
    return base


def nearest_prototypes_grouped_ok(feats, banks, k, distance_func: str = "cosine") -> bool:
    """``nearest_prototypes_grouped`` applies: cosine distance in a bf16 compute mode, f32 feature blocks of equal height that are
    consecutive rows of one buffer, at most 8 frozen-shape banks of one shape, and the one-product window search takes the shapes."""
    if distance_func != "cosine" or _state["compute"] == F32 or not 2 <= len(feats) <= 8 or len(banks) != len(feats):
        return False
    f0, b0 = feats[0], banks[0]
    if (f0.dtype != torch.float32 or not f0.is_cuda or any(f.shape != f0.shape for f in feats)
            or any(b.dtype != torch.float32 or b.shape != b0.shape or not b.is_contiguous() for b in banks)
            or not switches.enabled("grouped_search")):
        return False
    base = rows_of_one_buffer(feats)
    return base is not None and _window_search_ok(f0.shape[0], b0.shape[0], f0.shape[1], k, base, b0)


@torch.no_grad()
def nearest_prototypes_grouped(feats, banks, k, bank_norms):
    """([N, k] index lists of ``nearest_prototypes(feats[g], banks[g], k)`` for every g, the bf16 rounding of all feature rows
    [G * N, H]) as ONE chain of four launches -- row norms, rounding, a grouped bf16 product, a grouped window search
    (egk_topk_window_group) -- on the current stream: no fork, so the chain can sit on any stream of a capture (GraphONE.search_ahead),
    and the rounding is the activation-type copy the GraphONE stages consume.  Requires ``nearest_prototypes_grouped_ok``."""
    lib = _lib.load()
    base = rows_of_one_buffer(feats)
    G, (N, H) = len(feats), feats[0].shape
    K = banks[0].shape[0]
    # the screen's product from IEEE-half roundings (f16 matrix instructions: 11 significand bits, a window ~8 x narrower than
    # bf16's -- the prototype banks of a trained model put 30-90 prototypes inside the bf16 window of a row); values beyond the
    # half range make that row's window unbounded (slow, never wrong).  EGK_DISABLE=window_f16: the bf16 screen.
    f16 = _window_f16["on"]
    if base.is_contiguous() and H % 4 == 0 and switches.enabled("search_prep"):
        # row norms, the bf16 rounding and the half rounding in ONE pass over the rows (egk_row_inv_norm_cast: the bits of the three)
        f_norm = torch.empty(G * N, dtype=torch.float32, device=base.device)
        hi = torch.empty((G * N, H), dtype=torch.bfloat16, device=base.device)
        scr = torch.empty_like(hi) if f16 else hi
        _ck(lib.egk_row_inv_norm_cast(_stream(), _p(base), _p(f_norm), _p(hi), _p(scr) if f16 else None, G * N, H), "egk_row_inv_norm_cast")
    else:
        f_norm = row_inv_norm(base)
        hi = cast_raw(base, torch.bfloat16)
        scr = _cast_f16_bits(base) if f16 else hi
    ops_b = [_bank_window_operand(b, f16) for b in banks]
    dot = torch.empty((G * N, K), dtype=torch.float32, device=base.device)
    nn = torch.empty((G * N, k), dtype=torch.int64, device=base.device)
    gemm_grouped([((N, K, scr[g * N:(g + 1) * N], H, ops_b[g][0], H, H, dot[g * N:(g + 1) * N], K), {"compute": BF16, "op_f16": f16})
                  for g in range(G)])
    cand = _window_stats["cand"]
    if cand is not None and (cand.numel() != G * N or cand.device != base.device):
        cand = None
    dbg = switches.debug("window_cand") and not torch.cuda.is_current_stream_capturing()
    if dbg and cand is None:
        cand = torch.zeros(G * N, dtype=torch.int32, device=base.device)
    arr = lambda ts: (C.c_void_p * G)(*[t.data_ptr() for t in ts])
    _ck(lib.egk_topk_window_group16(_stream(), _p(dot), K, _p(base), base.stride(0), arr(banks), banks[0].stride(0), _p(f_norm),
                                    arr(bank_norms), arr([o[1] for o in ops_b]), _p(nn), _p(cand), G, N, K, H, k, int(f16)),
        "egk_topk_window_group16")
    if dbg:
        c = cand.view(G, N).float()
        print(f"[window_cand] K={K} k={k}: candidates per row mean {[round(float(v), 1) for v in c.mean(1)]} max {[int(v) for v in c.max(1).values]} "
              f"rb_max {[float(o[1]) for o in ops_b]}", flush=True)
    return [nn[g * N:(g + 1) * N] for g in range(G)], hi


def cosine_topk(f, bank, k, bank_inv_norm=None):
    return nearest_prototypes(f, bank, k, "cosine", bank_inv_norm)


@torch.no_grad()
def label_groups(label: torch.Tensor):
    """Group the rows with label >= 0 by label: (order int32 [n], seg_ptr int32 [G+1], seg_label int64 [G]) with the node
    ids of a group in ascending order (stable sort).  Host labels are grouped with numpy, device labels with a stable
    device sort -- integer work either way."""
    if label.device.type == "cpu":
        import numpy as np
        lb = label.numpy()
        keep = np.nonzero(lb >= 0)[0]
        order = keep[np.argsort(lb[keep], kind="stable")]
        srt = lb[order]
        starts = np.nonzero(np.concatenate([[True], srt[1:] != srt[:-1]]))[0] if srt.size else np.zeros(0, dtype=np.int64)
        seg_ptr = np.concatenate([starts, [srt.size]]).astype(np.int32)
        return (torch.from_numpy(order.astype(np.int32)), torch.from_numpy(seg_ptr),
                torch.from_numpy(srt[starts].astype(np.int64)))
    keep = torch.nonzero(label >= 0).reshape(-1)
    srt, perm = torch.sort(label[keep], stable=True)
    order = keep[perm]
    seg_label, counts = torch.unique_consecutive(srt, return_counts=True)
    seg_ptr = torch.zeros(seg_label.numel() + 1, dtype=torch.int32, device=label.device)
    seg_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order.to(torch.int32), seg_ptr, seg_label


@torch.no_grad()
def scatter_add_rows_f64(x, label, bank, count, groups=None):
    """bank[label[n]] += x[n] (fp64 bank, rows of one label summed in fp32 in node order first: what the reference's
    per-batch ``scatter(..., reduce='sum')`` does, graphone.py:53), count[label[n]] += 1; rows with label < 0 are skipped.
    Label-grouped segmented reduction, one wave per label: no atomics, bitwise reproducible.  ``groups`` =
    ``label_groups(label)`` when the caller already has it (one grouping per batch, several task banks)."""
    _need_gpu(x, bank)
    x = _c(x)
    order, seg_ptr, seg_label = groups if groups is not None else label_groups(label)
    dev = x.device
    if seg_label.numel() == 0:
        return  # no labelled row in this batch
    order, seg_ptr, seg_label = order.to(dev), seg_ptr.to(dev), seg_label.to(dev)
    _ck(_lib.load().egk_segment_sum_rows_f64(_stream(), _p(x), _p(order), _p(seg_ptr), _p(seg_label), _p(bank), _p(count),
                                            seg_label.numel(), x.shape[1], bank.shape[0], _dt(x)), "egk_segment_sum_rows_f64")


# ---- phase stamps (development) ---------------------------------------------------------------------------------
_stamps = {"buf": None, "names": []}


def stamps_enable(device="cuda", slots: int = 256):
    """Turn ``stamp(name)`` calls into one-lane launches that record the device wall clock (captured with the step:
    every replay rewrites the slots).  tools/phase_stamps.py reads them back."""
    _stamps["buf"] = torch.zeros(slots, dtype=torch.int64, device=device)
    _stamps["names"] = []


def stamp(name: str, seq: bool = False):
    """``seq``: the name gets a running index per step (reset by stamp("step_start")): stamps inside autograd nodes that
    run once per layer."""
    cb = _stamps.get("callbacks")
    if cb:
        k = _stamps.setdefault("cb_seq", {}).get(name, 0) if seq else None
        if seq:
            _stamps["cb_seq"][name] = k + 1
        fn = cb.get(f"{name}[{k}]" if seq else name)
        if fn is not None:
            fn()
    buf = _stamps["buf"]
    if buf is None:
        return
    if name == "step_start":
        _stamps["seq"] = {}
    if seq:
        k = _stamps.setdefault("seq", {}).get(name, 0)
        _stamps["seq"][name] = k + 1
        name = f"{name}[{k}]"
    names = _stamps["names"]
    if name in names:
        idx = names.index(name)
    else:
        names.append(name)
        idx = len(names) - 1
    _ck(_lib.load().egk_stamp(_stream(), _p(buf), idx), "egk_stamp")


def phase_callbacks(callbacks) -> None:
    """``{phase name: fn}`` (or None): ``fn()`` runs where the forward / backward code marks that phase with ``stamp(name)`` -- the
    engine records stream events at phases of one pass to order another pass behind them.  Sequence counters restart here."""
    _stamps["callbacks"] = dict(callbacks) if callbacks else None
    _stamps["cb_seq"] = {}


def stamps_read():
    """[(name, microseconds since the first stamp)] of the last recorded step (100 MHz counter)."""
    buf = _stamps["buf"]
    if buf is None:
        return []
    v = buf.cpu().tolist()
    t0 = min(v[i] for i in range(len(_stamps["names"])))
    return [(n, (v[i] - t0) / 100.0) for i, n in enumerate(_stamps["names"])]


# ---- profiling --------------------------------------------------------------------------------------------------
def prof_enable(on: bool):
    _lib.load().egk_prof_enable(int(on))


def prof_reset():
    _lib.load().egk_prof_reset()


def prof_report():
    """{kernel: dict(launches, total_ms, flops, bytes)} for kernels launched while profiling was on."""
    lib = _lib.load()
    out = {}
    name = C.create_string_buffer(64)
    n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
    for i in range(lib.egk_prof_count()):
        lib.egk_prof_get(i, name, 64, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by))
        if n.value:
            out[name.value.decode()] = dict(launches=n.value, total_ms=ms.value, flops=fl.value, bytes=by.value)
    return out
