"""Flat-buffer Adam: torch.optim.Adam semantics (L2 weight decay) in ONE kernel launch.

All trainable parameters that actually receive gradients are re-homed as views of one contiguous
fp32 buffer (``flat_p``); their ``.grad`` are views of ``flat_g``.  The backward kernels accumulate
weight gradients straight into those views (ops._grad_slot), ``zero_grad`` is one memset, the step
is one ``egk_adam_step`` launch and the data-parallel gradient exchange all-reduces slices of the
same buffer (dist.GradSync).  Mirrors ``_target_: torch.optim.Adam`` of configs/defaults.yaml:17-20.

Parameters whose gradient is None after the first backward (disabled tasks, detached aux heads,
frozen prototypes) are left alone, exactly as torch.optim.Adam skips ``grad is None``."""
from __future__ import annotations

import math
from typing import Iterable, List

import torch

from . import _lib, switches
from .ops import _ck, _p, _stream


def _minus_ranges(ranges, lo, hi):
    """``ranges`` ([a, b) pairs) without [lo, hi)."""
    out = []
    for a, b in ranges:
        if b <= lo or hi <= a:
            out.append((a, b))
            continue
        if a < lo:
            out.append((a, lo))
        if hi < b:
            out.append((hi, b))
    return out


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        params = [p for p in params]
        if len({id(p) for p in params}) != len(params):  # the reference passes some parameters twice
            seen, uniq = set(), []
            for p in params:
                if id(p) not in seen:
                    seen.add(id(p))
                    uniq.append(p)
            params = uniq
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat_p = self.flat_g = self.flat_m = self.flat_v = self.flat_w16 = None
        self.active: List[torch.Tensor] = []
        self.step_count = 0
        self.grad_scale = 1.0
        self._hyper = None
        self.flat_w16lo = None      # bf16 LOW halves of the parameters (p - bf16(p)), built on demand: ensure_lo_shadows
        self._lo_fresh = []         # [lo, hi) ranges of flat_w16lo that match flat_p (cleared by every write to flat_p)
        self._moment_views = {}     # id(param) -> (m view, v view) once materialised
        self._slot_of = {}          # id(param) -> (first element, slot length) in the flat buffers
        self._pending_state = None  # a state dict loaded before the flat buffers exist

    # -- construction of the flat buffers (first step, once the set of live gradients is known) -------
    def _materialise(self):
        group = self.param_groups[0]
        live = [p for p in group["params"] if p.requires_grad and p.grad is not None]
        if not live:
            raise RuntimeError("FlatAdam.step(): no parameter has a gradient")
        dev = live[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs parameters on a ROCm device (no CPU fallback)")
        # slots aligned to 16 bytes in the bf16 shadow (32 B in f32).  A matrix whose row count is not a multiple of
        # 64 (the classifier layers: 478, 115, 2 ... rows) gets its slot padded to whole 64-row blocks: the padding
        # stays zero under Adam (zero gradient, zero moments, zero weight), and the padded bf16 copy is the K-major
        # operand of the layer's dX contraction on the pipelined kernel (ops._Linear.backward).
        def slot(p):
            if p.dim() == 2 and p.shape[0] % 64:
                return (p.shape[0] + 63) // 64 * 64 * p.shape[1]
            if p.dim() == 1 and getattr(p, "_egk_bank", None) is not None:
                return (p.numel() + 63) // 64 * 64  # a bank's bias vector lines up with the padded rows of its weights
            return p.numel()
        live = self._bank_order(live)
        sizes = [(slot(p) + 7) // 8 * 8 for p in live]
        total = sum(sizes)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_w16 = torch.zeros(total, dtype=torch.bfloat16, device=dev)  # bf16 operand copies of the weights
        off = 0
        with torch.no_grad():
            for p, sz in zip(live, sizes):
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.data.reshape(-1))
                self.flat_g[off:off + n].copy_(p.grad.reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                p._egk_shadow = self.flat_w16[off:off + n].view(p.shape)
                self._moment_views[id(p)] = (self.flat_m[off:off + n].view(p.shape), self.flat_v[off:off + n].view(p.shape))
                self._slot_of[id(p)] = (off, sz)
                p._egk_lo_init = self.ensure_lo_shadows
                if p.dim() == 2 and p.shape[0] % 64 and p.shape[1] % 8 == 0:
                    rows64 = (p.shape[0] + 63) // 64 * 64
                    p._egk_shadow_rows64 = self.flat_w16[off:off + rows64 * p.shape[1]].view(rows64, p.shape[1])
                off += sz
        self.active = live
        self._bank_views(live)
        self.refresh_shadows()
        self._hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._hyper_src = torch.zeros(2, dtype=torch.float32, device=dev)  # {lr, grad_scale}: uploaded when they change
        self._t_dev = torch.zeros(1, dtype=torch.int64, device=dev)        # optimizer steps taken (device-side counter)
        self._src_host, self._t_mirror = None, 0
        if self._pending_state is not None:
            self._apply_state(self._pending_state)
            self._pending_state = None

    # -- classifier banks: several Linear layers over the SAME input (verb / noun classifiers of a head) ---------------
    # Their parameters carry ``_egk_bank = (owner, "w" | "b", index)`` (models/tasks/task.py).  The weights of a bank get
    # adjacent slots, and so do its biases: the zero-padded 64-row blocks of the members then form ONE [sum rows64, K]
    # matrix in the flat buffers (weights, bf16 copies, gradients), and the bank runs as one contraction forward, one for
    # dX, one for dW (ops.classifier_bank) instead of one of each per member.
    @staticmethod
    def _bank_order(live):
        groups, out, done = {}, [], set()
        for p in live:
            b = getattr(p, "_egk_bank", None)
            if b is not None:
                groups.setdefault((b[0], b[1]), []).append((b[2], p))
        for p in live:
            b = getattr(p, "_egk_bank", None)
            if b is None:
                out.append(p)
            elif (b[0], b[1]) not in done:
                done.add((b[0], b[1]))
                out.extend(q for _, q in sorted(groups[(b[0], b[1])], key=lambda t: t[0]))
        return out

    def _bank_views(self, live):
        banks = {}
        for p in live:
            b = getattr(p, "_egk_bank", None)
            if b is not None:
                banks.setdefault(b[0], {}).setdefault(b[1], []).append((b[2], p))
                p._egk_bank_views = None
        for kinds in banks.values():
            ws = [q for _, q in sorted(kinds.get("w", []), key=lambda t: t[0])]
            bs = [q for _, q in sorted(kinds.get("b", []), key=lambda t: t[0])]
            if len(ws) < 2 or len(bs) != len(ws) or len({w.shape[1] for w in ws}) != 1 or ws[0].shape[1] % 64:
                continue
            cols = ws[0].shape[1]
            rows64 = [(w.shape[0] + 63) // 64 * 64 for w in ws]
            offs_w, offs_b = [self._slot_of[id(w)][0] for w in ws], [self._slot_of[id(b)][0] for b in bs]
            ok = all(offs_w[i + 1] == offs_w[i] + rows64[i] * cols for i in range(len(ws) - 1))
            ok = ok and all(offs_b[i + 1] == offs_b[i] + rows64[i] for i in range(len(bs) - 1))
            ok = ok and all(b.numel() == w.shape[0] for w, b in zip(ws, bs))
            if not ok:
                continue
            n, ow, ob = sum(rows64), offs_w[0], offs_b[0]
            starts = [sum(rows64[:i]) for i in range(len(ws))]
            ws[0]._egk_bank_views = {
                "rows": [(st, w.shape[0]) for st, w in zip(starts, ws)], "n": n, "k": cols,
                "w16": self.flat_w16[ow:ow + n * cols].view(n, cols), "wp": self.flat_p[ow:ow + n * cols].view(n, cols),
                "wg": self.flat_g[ow:ow + n * cols].view(n, cols), "b": self.flat_p[ob:ob + n], "bg": self.flat_g[ob:ob + n]}

    def region_of(self, params) -> tuple:
        """[lo, hi) of the flat buffers spanned by ``params`` (those that live there); (0, 0) if none does."""
        slots = [self._slot_of[id(p)] for p in params if id(p) in self._slot_of]
        if not slots:
            return (0, 0)
        return (min(o for o, _ in slots), max(o + n for o, n in slots))

    # -- checkpointing: the layout of torch.optim.Adam's state dict (per-parameter exp_avg / exp_avg_sq / step) -----
    def state_dict(self):
        """``{"state": {index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}`` with parameter indices
        in constructor order -- loadable by torch.optim.Adam over the same parameter list and vice versa."""
        if getattr(self, "_moments_sharded", False):
            raise RuntimeError("FlatAdam.state_dict(): the moments are sharded over the ranks (dist.GradSync shard_update: every rank "
                               "holds its own 1 / world slice, zeros elsewhere) -- call GradSync.gather_moments(optimizer) on "
                               "EVERY rank before saving")
        group = self.param_groups[0]
        state = {}
        for i, p in enumerate(group["params"]):
            mv = self._moment_views.get(id(p))
            if mv is not None:
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": mv[0].detach().clone(),
                            "exp_avg_sq": mv[1].detach().clone()}
        if self._pending_state is not None and not state:
            return self._pending_state
        pg = {k: v for k, v in group.items() if k != "params"}
        pg["params"] = list(range(len(group["params"])))
        return {"state": state, "param_groups": [pg]}

    def load_state_dict(self, state_dict):
        """Hyper-parameters now; moments now if the flat buffers exist, otherwise when the first step builds them."""
        pg = state_dict["param_groups"][0]
        for k, v in pg.items():
            if k != "params" and k in self.param_groups[0]:
                self.param_groups[0][k] = v
        if self.materialised:
            self._apply_state(state_dict)
        else:  # snapshot: the caller may keep using (or another optimizer may step) the tensors it handed in
            self._pending_state = {"state": {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                                             for i, st in state_dict["state"].items()},
                                   "param_groups": state_dict["param_groups"]}
            steps = [float(s["step"]) for s in state_dict["state"].values() if "step" in s]
            self.step_count = int(max(steps)) if steps else 0
            # The parameters with saved moments ARE the live set of the run that wrote the state: build the flat buffers
            # now, so that the first step after a resume runs on the layout (bf16 operand copies, classifier banks) every
            # later step of the interrupted run ran on -- a resumed run continues it bit for bit.
            params = self.param_groups[0]["params"]
            live = [params[int(i)] for i in state_dict["state"] if int(i) < len(params)]
            if live and all(p.is_cuda and p.requires_grad for p in live):
                for p in live:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._materialise()

    def _apply_state(self, state_dict):
        params = self.param_groups[0]["params"]
        steps = []
        with torch.no_grad():
            for i, st in state_dict["state"].items():
                mv = self._moment_views.get(id(params[int(i)]))
                if mv is None:
                    continue  # saved for a parameter that receives no gradient in this run
                mv[0].copy_(st["exp_avg"])
                mv[1].copy_(st["exp_avg_sq"])
                steps.append(float(st["step"]))
        if steps:
            self.step_count = int(max(steps))

    # -- low halves for the three-product ('bf16x3') contractions ---------------------------------------------------------
    # A weight enters such a contraction as hi + lo with hi = its bf16 shadow (kept by the Adam kernel) and lo = bf16(p - hi).
    # The lo buffer is allocated the first time it is asked for; ``refresh_lo_shadows(params)`` recomputes the region spanned
    # by ``params`` in ONE launch (the engine: the backbone's region at the start of the forward-only precise pass of a step),
    # and a parameter whose slot is not marked fresh when a contraction wants its lo half refreshes its own slot
    # (ops._x3_weight) -- correctness never depends on the caller having refreshed.
    def ensure_lo_shadows(self):
        if not self.materialised:
            return False
        if self.flat_w16lo is None:
            self.flat_w16lo = torch.zeros_like(self.flat_w16)
            self._lo_fresh = []
            for p in self.active:
                off, _ = self._slot_of[id(p)]
                n = p.numel()
                view = self.flat_w16lo[off:off + n].view(p.shape)
                p._egk_lo = (view, (lambda o=off, m=n: self._lo_is_fresh(o, m)), (lambda q=p: self.refresh_lo_shadows([q])))
        return True

    # Once the low halves exist the Adam launch keeps them: it writes bf16(p - bf16(p)) of the slice it updates beside the bf16
    # shadow (+ 2 B per parameter on a 30 B pass), so the precise pass at the head of the NEXT step finds them fresh instead of
    # splitting the backbone's 17 M parameters in a launch of its own at the head of the step's critical chain (38 us in BASELINE
    # config 4, profiles/r04_c4_replay_timeline.txt at 118 us).  EGK_DISABLE=adam_lo: the round-4 behaviour.
    adam_writes_lo = switches.enabled("adam_lo")

    def _lo_is_fresh(self, off: int, n: int) -> bool:
        # (fresh ranges may have been cut by partial updates: a slot is fresh when the union of the ranges covers it)
        need = [(off, off + n)]
        for lo, hi in self._lo_fresh:
            need = [piece for a, b in need for piece in ((a, min(b, lo)), (max(a, hi), b)) if piece[1] > piece[0]]
            if not need:
                return True
        return not need

    def invalidate_lo_shadows(self):
        self._lo_fresh = []

    def refresh_lo_shadows(self, params=None):
        """flat_w16lo[lo:hi] = bf16(flat_p - bf16(flat_p)) over the region spanned by ``params`` (default: everything)."""
        if not self.ensure_lo_shadows():
            return
        lo, hi = (0, self.flat_p.numel()) if params is None else self.region_of(list(params))
        if hi <= lo or self._lo_is_fresh(lo, hi - lo):
            return
        _ck(_lib.load().egk_split_bf16(_stream(), _p(self.flat_p[lo:hi]), hi - lo, None, _p(self.flat_w16lo[lo:hi]), hi - lo, 1, hi - lo),
            "egk_split_bf16")
        self._lo_fresh.append((lo, hi))

    def refresh_shadows(self, lo: int = 0, hi=None):
        """Re-derive the bf16 operand copies of flat_p[lo:hi] (default: everything) from the f32 parameters: after
        load_state_dict, the all-gather of a sharded update, or any other write to the parameters that did not go through ``step``."""
        hi = self.flat_p.numel() if hi is None else hi
        if hi <= lo:
            return
        if self.flat_w16 is not None:
            _ck(_lib.load().egk_cast(_stream(), _p(self.flat_p[lo:hi]), 0, _p(self.flat_w16[lo:hi]), 1, hi - lo), "egk_cast")
        self._lo_fresh = _minus_ranges(self._lo_fresh, lo, hi)
        if self.flat_w16lo is not None and self.adam_writes_lo:
            # a captured step whose Adam launches keep the low halves holds NO split launch (engine.StepBase.capture): an
            # out-of-band write to the parameters (checkpoint load, a restored snapshot, gathered slices) leaves them fresh itself
            _ck(_lib.load().egk_split_bf16(_stream(), _p(self.flat_p[lo:hi]), hi - lo, None, _p(self.flat_w16lo[lo:hi]), hi - lo, 1, hi - lo),
                "egk_split_bf16")
            self._lo_fresh.append((lo, hi))

    @property
    def materialised(self) -> bool:
        return self.flat_p is not None

    def zero_grad(self, set_to_none: bool = False):
        if not self.materialised:
            return super().zero_grad(set_to_none=True)
        self.zero_flat_grads()

    def zero_flat_grads(self) -> None:
        """The flat gradient buffer cleared by one launch of the library (capturable)."""
        g = self.flat_g
        nbytes = g.numel() * g.element_size()
        if nbytes % 16 == 0 and g.data_ptr() % 16 == 0 and self._zero_all_but_stored():
            return
        if nbytes % 16 or g.data_ptr() % 16:
            g.zero_()
            return
        _ck(_lib.load().egk_zero_fill(_stream(), _p(g), nbytes), "egk_zero_fill")

    # The step constants {lr, 1-b1^t, sqrt(1-b2^t), grad_scale} are computed ON THE DEVICE (egk_adam_hyper) from a device-side
    # step counter and a two-float source {lr, grad_scale}: the launch is a node of a captured step, so a replay needs no
    # host -> device copy in front of it (4 us of copy + the gap behind it, 14 us per step of the headline workload) while lr
    # schedules and step counts keep advancing.  The host uploads the source only when lr / grad_scale change (per epoch) and
    # the counter only when ``step_count`` was set from outside (load_state_dict); ``_t_mirror`` is the value the device
    # counter will hold once everything enqueued so far has run.
    def sync_hyper_source(self) -> None:
        src = (float(self.param_groups[0]["lr"]), float(self.grad_scale))
        if src != self._src_host:
            # (a FRESH pinned staging buffer per upload: torch's caching host allocator does not hand a pinned block out again
            #  before the async copy that read it has completed)
            host = torch.empty(2, dtype=torch.float32, pin_memory=True)
            host[0], host[1] = src
            self._hyper_src.copy_(host, non_blocking=True)
            self._src_host = src
        if self._t_mirror != self.step_count:
            host = torch.empty(1, dtype=torch.int64, pin_memory=True)
            host[0] = self.step_count
            self._t_dev.copy_(host, non_blocking=True)
            self._t_mirror = self.step_count

    def prepare_hyper(self, in_capture: bool = False):
        """The constants of the NEXT step (t = step_count + 1), on the current stream.  ``in_capture``: the launch is being
        recorded into a graph -- the caller has called ``sync_hyper_source`` before the capture and calls
        ``note_captured_step`` after every replay."""
        b1, b2 = self.param_groups[0]["betas"]
        if not in_capture:
            self.sync_hyper_source()
        _ck(_lib.load().egk_adam_hyper(_stream(), _p(self._hyper_src), _p(self._t_dev), float(b1), float(b2), _p(self._hyper)),
            "egk_adam_hyper")
        if not in_capture:
            self._t_mirror += 1

    def note_captured_step(self) -> None:
        """A replayed graph that contains the constants launch has been enqueued: the device counter moves on with it."""
        self._t_mirror += 1

    # -- gradient slots with ONE writer per step: stored, not accumulated -----------------------------------------------------------
    # The flat gradient buffer is cleared every step (100-260 MB beside the forward pass) so that the weight-gradient launches can
    # ADD into it -- which also makes each of them read its zeros back.  A weight matrix whose gradient comes from exactly one
    # launch per step needs neither: that launch stores.  Which slots those are is LEARNT from an eager step (every dW-form launch
    # into a parameter's slot is counted; a slot counted once qualifies) and CHECKED in the capture: a stored slot written twice,
    # or not at all, is an error -- never a silently wrong gradient.  Every capture: one rank, the staged graphs, the one-graph exchange.
    def _matrix_index(self):
        idx = getattr(self, "_mat_index", None)
        if idx is None:
            idx = self._mat_index = {self._slot_of[id(p)][0]: tuple(p.shape) for p in self.active
                                     if p.dim() == 2 and getattr(p, "_egk_bank", None) is None}
        return idx

    def learn_begin(self):
        """Provider for ops.set_grad_slot_provider during an EAGER step: counts the launches per matrix slot, changes nothing."""
        idx, g0, counts = self._matrix_index(), self.flat_g.data_ptr(), {}
        self._learn_counts = counts

        def provider(out, M, N, ldc):
            d = out.data_ptr() - g0
            if d >= 0 and d % 4 == 0 and ldc == N and idx.get(d // 4) == (M, N):
                counts[d // 4] = counts.get(d // 4, 0) + 1
            return None
        return provider

    def learn_end(self):
        idx = self._matrix_index()
        self.store_slots = {off: idx[off][0] * idx[off][1] for off, c in (getattr(self, "_learn_counts", None) or {}).items() if c == 1}
        self._learn_counts = None

    def store_begin(self):
        """Provider for a CAPTURE: 'store' for the learnt single-writer slots; ``zero_flat_grads`` leaves them out meanwhile."""
        slots = getattr(self, "store_slots", None)
        if not slots:
            return None
        idx, g0 = self._matrix_index(), self.flat_g.data_ptr()
        self._store_claims, self._store_active = set(), True

        def provider(out, M, N, ldc):
            d = out.data_ptr() - g0
            if d < 0 or d % 4 or ldc != N or (d // 4) not in slots or idx.get(d // 4) != (M, N):
                return None
            if d // 4 in self._store_claims:
                raise RuntimeError("grad_store: a gradient slot learnt as written once per step is written twice in the captured step "
                                   "(set EGK_DISABLE=grad_store)")
            self._store_claims.add(d // 4)
            return "store"
        return provider

    def store_end(self, ok: bool = True):
        claims, self._store_active = getattr(self, "_store_claims", None), False
        self._store_claims = None
        if ok and claims is not None and claims != set(self.store_slots):
            raise RuntimeError(f"grad_store: {len(set(self.store_slots) - claims)} gradient slot(s) left uncleared were not written by the "
                               "captured step (set EGK_DISABLE=grad_store)")

    def _zero_all_but_stored(self) -> bool:
        if not getattr(self, "_store_active", False):
            return False
        import ctypes as C
        total, at, rest = self.flat_g.numel(), 0, []
        for off in sorted(self.store_slots):
            n = self.store_slots[off]
            a, b = (off + 3) // 4 * 4, (off + n) // 4 * 4  # (whole 16-byte groups INSIDE the slot stay uncleared; its ragged ends are cleared)
            if b <= a:
                continue
            if a > at:
                rest.append((at, a - at))
            at = b
        if total > at:
            rest.append((at, total - at))
        for i in range(0, len(rest), 48):
            part = rest[i:i + 48]
            bg, ln = (C.c_int64 * len(part))(*[4 * b for b, _ in part]), (C.c_int64 * len(part))(*[4 * n for _, n in part])
            _ck(_lib.load().egk_zero_fill_ranges(_stream(), _p(self.flat_g), bg, ln, len(part)), "egk_zero_fill_ranges")
        return True

    def launch(self, grads=None, lo: int = 0, hi=None, bump=None):
        """The kernel launch alone (capturable).  ``grads``: the buffer to read gradients from (default the f32
        flat buffer; dist.GradSync hands in its bf16 copy after a compressed all-reduce).  ``[lo, hi)``: element range
        of the flat buffers to update (multiples of 8; the pipelined gradient exchange steps chunk by chunk).
        ``bump`` = (int64 device word, stride): the word moves on by ``stride`` inside this launch (egk_adam_step_bump)."""
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        grads = self.flat_g if grads is None else grads
        hi = self.flat_p.numel() if hi is None else hi
        if hi <= lo:
            return
        sl = slice(lo, hi)
        lo16 = self.flat_w16lo[sl] if (self.flat_w16lo is not None and self.adam_writes_lo) else None
        if lo16 is not None:
            # the launch also writes the low halves of the slice it updates (egk_adam_step_bump): that slice is fresh, the rest as it was
            self._lo_fresh = _minus_ranges(self._lo_fresh, lo, hi) + [(lo, hi)]
        elif self._lo_fresh:
            self._lo_fresh = []  # (the parameters move: every low half is stale)
        if bump is not None or lo16 is not None:
            _ck(_lib.load().egk_adam_step_bump(_stream(), _p(self.flat_p[sl]), _p(grads[sl]), 1 if grads.dtype == torch.bfloat16 else 0,
                                               _p(self.flat_m[sl]), _p(self.flat_v[sl]), hi - lo, _p(self._hyper), b1, b2,
                                               g["eps"], g["weight_decay"], _p(self.flat_w16[sl]), _p(lo16),
                                               _p(bump[0]) if bump is not None else None, int(bump[1]) if bump is not None else 0),
                "egk_adam_step_bump")
            return
        _ck(_lib.load().egk_adam_step(_stream(), _p(self.flat_p[sl]), _p(grads[sl]), 1 if grads.dtype == torch.bfloat16 else 0,
                                      _p(self.flat_m[sl]), _p(self.flat_v[sl]), hi - lo, _p(self._hyper), b1, b2,
                                      g["eps"], g["weight_decay"], _p(self.flat_w16[sl])),
            "egk_adam_step")

    @torch.no_grad()
    def step(self, closure=None, grads=None):
        if not self.materialised:
            self._materialise()
        self.prepare_hyper()
        self.launch(grads)
        self.step_count += 1
