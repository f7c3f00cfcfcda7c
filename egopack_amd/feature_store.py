"""Device-resident clip-feature store and the reference's segment sampling on top of it.

The reference keeps one ``[frames, F]`` array per video on disk (``np.load(..., mmap_mode='r')``,
data/ego4d_fho.py:97-105) and builds every sample on the CPU: per action window, S frame indices
(``BaseFrameDataset.random_sampling_indices`` / ``uniform_sampling_indices``, data/base_dataset.py:128-155), then
``np.take`` of those rows; the resulting ``[T, S, F]`` block travels to the GPU every step (18 KB per node in f32).

MI355X-first: 288 GB of HBM hold the whole store (Ego4D FHO at stride 16: ~25 M rows x 1536 = 77 GB in bf16), so
    * ``FeatureStore`` keeps all videos as ONE ``[rows, F]`` device tensor + a ``video -> (first row, length)`` map;
    * the index arithmetic stays on the host, on the same numpy ``RandomState`` stream the reference consumes
      (integer work: bit-exact, pinned by tests/golden/sampling.pt);
    * a batch carries ``x_idx [N, S]`` (global row ids, -1 = the reference's all-zero clip) instead of ``x``; 8 bytes per
      (node, segment) cross PCIe instead of 6 KB, and ``egk_gather_rows`` materialises ``x`` in HBM
      (``FeatureStore.gather``; the engine gathers straight into the packed multi-task buffer).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops


# ---- index arithmetic (host, numpy semantics of the reference) ----------------------------------------------------
def random_sampling_indices(size: int, n: int, rng=np.random) -> np.ndarray:
    """data/base_dataset.py:128-139 (consumes ``rng.randint(average_duration, size=n)`` exactly when size >= n)."""
    average_duration = size // n
    if average_duration > 0:
        indices = np.multiply(list(range(n)), size / n)
        indices = indices + rng.randint(average_duration, size=n)
        indices = np.clip(indices, 0, size)
    else:
        indices = np.linspace(0, size, n, endpoint=False, dtype=int)
    return np.round(indices).astype(int)


def uniform_sampling_indices(size: int, n: int) -> np.ndarray:
    """data/base_dataset.py:141-145."""
    offsets = np.linspace(0, size, n, endpoint=False, dtype=int)
    return (offsets + (size // n // 2)).astype(int)


def window_rows(first_row: int, video_len: int, start: int, end: int, n: int, random: bool, rng=np.random) -> np.ndarray:
    """Global store rows of ``sampling(video_features[start:end], n)`` (ego4d_fho.py:228-236 and :376-386).
    Mirrors numpy slicing + ``np.take``: the slice clips to the video, an index past the slice (random sampling may
    produce ``size`` itself) or an empty window raises in the reference, whose ``except`` substitutes an all-zero
    clip -> n times -1 here."""
    lo, hi = min(max(int(start), 0), video_len), min(max(int(end), 0), video_len)  # numpy slice clipping (start, end >= 0)
    size = max(hi - lo, 0)
    if size == 0:  # empty slice: size // n == 0 -> the linspace branch (no random numbers drawn), np.take raises -> zeros
        return np.full(n, -1, dtype=np.int64)
    idx = random_sampling_indices(size, n, rng) if random else uniform_sampling_indices(size, n)
    if idx.max(initial=0) >= size or idx.min(initial=0) < -size:
        return np.full(n, -1, dtype=np.int64)  # np.take raises IndexError -> zero clip
    idx = np.where(idx < 0, idx + size, idx)
    return (first_row + lo + idx).astype(np.int64)


_mt_direct = {}  # id(bit generator) -> (bit generator, key array, position cell) viewing its state in place


def _mt_state(rng):
    """(state tuple or None, key array, position cell) of a legacy RandomState / the numpy.random module, for the host
    helpers that advance the Mersenne Twister in place; ``_mt_commit`` writes the advanced state back.

    ``get_state`` + ``set_state`` cost ~70 us per call pair -- more than sampling 2048 windows.  numpy's MT19937 bit
    generator publishes the address of its state (``ctypes.state_address`` -> ``struct { uint32_t key[624]; int pos; }``), so
    the helpers run on the generator's own memory; the layout is CHECKED against ``get_state()`` the first time a generator is
    seen, and a generator that does not pass takes the get_state / set_state route."""
    owner = getattr(rng, "_bit_generator", None)
    if owner is None and hasattr(rng, "mtrand"):  # the numpy.random module: its global RandomState
        owner = getattr(getattr(rng.mtrand, "_rand", None), "_bit_generator", None)
    hit = _mt_direct.get(id(owner)) if owner is not None else None
    if hit is not None and hit[0] is owner:
        return None, hit[1], hit[2]
    state = rng.get_state()
    if state[0] != "MT19937":
        raise ValueError(f"the batch builders replay numpy's legacy MT19937 stream, not {state[0]}")
    if owner is not None and id(owner) not in _mt_direct:
        try:
            import ctypes
            addr = int(owner.ctypes.state_address)
            key = np.ctypeslib.as_array((ctypes.c_uint32 * 624).from_address(addr))
            pos = np.ctypeslib.as_array((ctypes.c_int32 * 1).from_address(addr + 624 * 4))
            ok = np.array_equal(key, state[1]) and int(pos[0]) == int(state[2])
        except Exception:
            ok = False
        if len(_mt_direct) >= 64:  # (the entries keep their generators alive: bounded)
            _mt_direct.clear()
        _mt_direct[id(owner)] = (owner, key, pos) if ok else (None, None, None)
        if ok:
            return None, key, pos
    key = np.ascontiguousarray(state[1], dtype=np.uint32)
    return state, key, np.array([state[2]], dtype=np.int32)


def _mt_commit(rng, state, key, pos) -> None:
    if state is not None:  # (None: the helpers advanced the generator's own memory)
        rng.set_state((state[0], key, int(pos[0]), *state[3:]))


def randint_sequence(rng, highs: np.ndarray, n: int) -> np.ndarray:
    """``np.stack([rng.randint(h, size=n) for h in highs])`` (h <= 0: a row of zeros, nothing drawn) on the SAME random
    stream, without the per-window call: the library's host helper ``egk_host_bounded_draws`` runs the Mersenne Twister of
    ``rng`` itself (state taken with ``get_state``, advanced in C by numpy's masked-rejection sampling bound by bound, written
    back with ``set_state``), so ``rng`` is left exactly where the per-window calls would have left it.
    ``rng``: a ``numpy.random.RandomState`` or the ``numpy.random`` module (the reference uses the module-level stream)."""
    highs = np.ascontiguousarray(highs, dtype=np.int64)
    W = int(highs.shape[0])
    out = np.zeros((W, n), dtype=np.int64)
    if not (highs > 1).any():
        return out
    state, key, pos = _mt_state(rng)
    rc = int(_lib.load().egk_host_bounded_draws(key.ctypes.data, pos.ctypes.data, highs.ctypes.data, W, n, out.ctypes.data))
    if rc != 0:
        raise ValueError(f"randint_sequence: unsupported bound (code {rc})")
    _mt_commit(rng, state, key, pos)
    return out


def window_rows_batch(first_row, video_len, start, end, n: int, random: bool, rng=np.random, native: bool = True) -> np.ndarray:
    """``np.stack([window_rows(first_row[w], video_len[w], start[w], end[w], n, random, rng) for w in range(W)])`` for all W
    windows at once, consuming ``rng`` exactly as the W calls would: [W, n] int64 store rows, a window the reference replaces by
    an all-zero clip is a row of -1.  ``native``: the library's host helper ``egk_host_window_rows`` (one C loop over the
    windows, ~30 us per 2048 windows); False: the same thing as numpy vector arithmetic (~0.6 ms; the two are tested equal)."""
    first_row, video_len = np.ascontiguousarray(first_row, dtype=np.int64), np.ascontiguousarray(video_len, dtype=np.int64)
    start, end = np.ascontiguousarray(start, dtype=np.int64), np.ascontiguousarray(end, dtype=np.int64)
    if native:
        W = int(start.shape[0])
        out = np.empty((W, n), dtype=np.int64)
        lib = _lib.load()
        args = (first_row.ctypes.data, video_len.ctypes.data, start.ctypes.data, end.ctypes.data, W, n)
        if not random:
            rc = int(lib.egk_host_window_rows(None, None, *args, 0, out.ctypes.data))
        else:
            state, key, pos = _mt_state(rng)
            rc = int(lib.egk_host_window_rows(key.ctypes.data, pos.ctypes.data, *args, 1, out.ctypes.data))
            if rc == 0:
                _mt_commit(rng, state, key, pos)
        if rc != 0:
            raise ValueError(f"window_rows_batch: host helper failed (code {rc})")
        return out
    lo = np.minimum(np.maximum(start, 0), video_len)
    hi = np.minimum(np.maximum(end, 0), video_len)
    size = np.maximum(hi - lo, 0)
    W = size.shape[0]
    k = np.arange(n, dtype=np.float64)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        lin = np.floor(k * (size[:, None] / n)).astype(np.int64)  # np.linspace(0, size, n, endpoint=False, dtype=int)
    if random:
        avg = size // n
        draws = randint_sequence(rng, np.where(size > 0, avg, 0), n)  # (windows with avg == 0 or size == 0 draw nothing)
        idx_r = np.round(np.clip(k * (size[:, None] / n) + draws, 0, size[:, None])).astype(np.int64)
        idx = np.where((avg > 0)[:, None], idx_r, lin)
    else:
        idx = lin + (size // n // 2)[:, None]
    bad = (size == 0) | (idx.max(axis=1, initial=0) >= size)
    rows = first_row[:, None] + lo[:, None] + idx
    return np.where(bad[:, None], np.int64(-1), rows).astype(np.int64).reshape(W, n)


# ---- the store --------------------------------------------------------------------------------------------------------
class FeatureStore:
    def __init__(self, videos: Mapping[str, np.ndarray], device="cuda", dtype: torch.dtype = torch.bfloat16,
                 chunk_rows: int = 1 << 16):
        """``videos``: uid -> [frames, F] array (numpy array or ``np.load(..., mmap_mode='r')`` memmap)."""
        self.offsets: Dict[str, Tuple[int, int]] = {}
        total, feat = 0, None
        for uid, arr in videos.items():
            if feat is None:
                feat = int(arr.shape[1])
            elif int(arr.shape[1]) != feat:
                raise ValueError(f"video {uid}: feature size {arr.shape[1]} != {feat}")
            self.offsets[uid] = (total, int(arr.shape[0]))
            total += int(arr.shape[0])
        if feat is None:
            raise ValueError("FeatureStore: no videos")
        self.features_size, self.rows = feat, total
        self.table = torch.empty((total, feat), dtype=dtype, device=device)
        for uid, arr in videos.items():  # chunked upload: a memmap is never materialised in one piece
            first, n = self.offsets[uid]
            for r in range(0, n, chunk_rows):
                blk = torch.from_numpy(np.ascontiguousarray(arr[r:r + chunk_rows]))
                self.table[first + r: first + r + blk.shape[0]].copy_(blk.to(device, non_blocking=False))

    @classmethod
    def from_npy_dir(cls, path, video_uids: Sequence[str], **kw) -> "FeatureStore":
        """The reference's processed layout: ``<path>/<video_uid>.npy`` (ego4d_fho.py:97-105)."""
        path = Path(path)
        return cls({uid: np.load(path / f"{uid}.npy", mmap_mode="r") for uid in video_uids}, **kw)

    def gather(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None, dtype: Optional[torch.dtype] = None,
               hi: Optional[torch.Tensor] = None, w: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x[..., :] = table[idx[...], :] (zeros where idx < 0): ``[N, S] -> [N, S, F]`` on the device.  With ``hi`` / ``w``
        (same shape; w float64): the interpolating form ``egk_gather_lerp_rows`` (PNR sampling, LTA 'avg' nodes)."""
        dev = self.table.device
        idx = idx.to(dev, dtype=torch.int64, non_blocking=True).contiguous()
        shape = (*idx.shape, self.features_size)
        if out is None:
            out = torch.empty(shape, dtype=dtype or self.table.dtype, device=dev)
        elif tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError(f"FeatureStore.gather: out must be contiguous {shape}")
        ops._need_gpu(self.table, out)
        n = idx.numel()
        if not n:
            return out
        lib = _lib.load()
        if hi is None:
            ops._ck(lib.egk_gather_rows(ops._stream(), ops._p(self.table), ops._dt(self.table), self.table.stride(0), self.rows,
                                        ops._p(idx), ops._p(out), ops._dt(out), n, self.features_size), "egk_gather_rows")
        else:
            hi = hi.to(dev, dtype=torch.int64, non_blocking=True).contiguous()
            w = w.to(dev, dtype=torch.float64, non_blocking=True).contiguous()
            if hi.shape != idx.shape or w.shape != idx.shape:
                raise ValueError("FeatureStore.gather: idx / hi / w shapes differ")
            ops._ck(lib.egk_gather_lerp_rows(ops._stream(), ops._p(self.table), ops._dt(self.table), self.table.stride(0), self.rows,
                                             ops._p(idx), ops._p(hi), ops._p(w), ops._p(out), ops._dt(out), n, self.features_size),
                    "egk_gather_lerp_rows")
        return out

    def gather_item(self, item, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """Feature block of one sample built by ``ar_item`` / ``lta_item`` / ``oscc_item`` / ``pnr_item``."""
        lo = torch.from_numpy(np.ascontiguousarray(item["lo"]))
        if "hi" in item:
            return self.gather(lo, dtype=dtype, hi=torch.from_numpy(np.ascontiguousarray(item["hi"])),
                               w=torch.from_numpy(np.ascontiguousarray(item["w"], dtype=np.float64)))
        return self.gather(lo, dtype=dtype)


def materialise_features(batch, store: FeatureStore, dtype: Optional[torch.dtype] = None):
    """Give a batch that carries ``x_idx`` its ``x`` (no-op when ``x`` is already there)."""
    if getattr(batch, "x", None) is None and getattr(batch, "x_idx", None) is not None:
        batch.x = store.gather(batch.x_idx, dtype=dtype)
    return batch


# ---- per-task sample builders: the index half of the reference datasets' ``get`` ------------------------------------
# Each returns the sample as INDEX matrices over the resident store -- ``lo`` [T, S] (and, where the reference
# interpolates or averages, ``hi`` [T, S] + ``w`` [T, S]: x = table[lo] if lo == hi else (1 - w) table[lo] + w table[hi]) --
# plus labels / positions / scalar attributes exactly as the reference builds them.  -1 = the reference's all-zero clip.
def ar_item(first_row: int, video_len: int, actions, window_size: int, stride: int, S: int, random: bool, rng=np.random):
    """Ego4dRecognitionDataset.get (ego4d_fho.py:217-242).  ``actions``: the window's (start_frame, end_frame, verb, noun)."""
    T = len(actions)
    c = window_size // 2
    y = np.full((T, 2), -1, dtype=np.int64)
    y[c] = (actions[c][2], actions[c][3])
    pos = np.arange(T, dtype=np.int64) - c
    rows = np.stack([window_rows(first_row, video_len, a[0] // stride, min(video_len - 1, a[1] // stride), S, random, rng)
                     for a in actions])
    return {"lo": rows, "y": y, "pos": pos}


def lta_item(first_row: int, video_len: int, input_clips, forecast_labels, n_forecast: int, stride: int, S: int, random: bool,
             append_node: str = "avg", rng=np.random):
    """Ego4dLTADataset.get (ego4d_fho.py:353-396): input clips sampled like AR windows (start shifted by one stride
    unit), forecast nodes = mean of the input clips ('avg': two input clips -> a w = 1/2 interpolation), zeros
    ('zero'), or uniform noise ('random': not index-expressible, unsupported)."""
    n_in = len(input_clips)
    rows = [window_rows(first_row, video_len, max(1, a[0] // stride) - 1, min(video_len - 1, a[1] // stride), S, random, rng)
            for a in input_clips]
    lo, hi, w = list(rows), list(rows), [np.zeros(S)] * n_in
    if append_node == "avg":
        if n_in != 2:
            raise NotImplementedError("lta_item: the 'avg' forecast nodes are built for 2 input clips (the reference default)")
        for _ in range(n_forecast):
            lo.append(rows[0]); hi.append(rows[1]); w.append(np.full(S, 0.5))
    elif append_node == "zero":
        for _ in range(n_forecast):
            lo.append(np.full(S, -1, dtype=np.int64)); hi.append(np.full(S, -1, dtype=np.int64)); w.append(np.zeros(S))
    else:
        raise NotImplementedError(f"lta_item: append_node={append_node!r}")
    y = np.full((n_in + n_forecast, 2), -1, dtype=np.int64)
    y[n_in:] = np.asarray(forecast_labels, dtype=np.int64).reshape(n_forecast, 2)
    return {"lo": np.stack(lo), "hi": np.stack(hi), "w": np.stack(w), "y": y, "pos": np.arange(n_in + n_forecast, dtype=np.int64)}


def oscc_item(first_row: int, video_len: int, start_frame: int, end_frame: int, pnr_frame, state_change: int, stride: int, S: int,
              train: bool, aug_prob: float = 0.0, rng=np.random, py_random=None):
    """Ego4dOSCCDataset.get (ego4d_oscc.py:191-223): 4 * S segments drawn from the clip (sorted random choice on
    train, linspace otherwise), reshaped to 4 nodes; on train, with probability ``aug_prob``, a state-change clip is
    the reference tries a frame-replacement augmentation that is not executable as written (see below; aug_prob is 0 in
    the experiments)."""
    import random as _random
    py_random = py_random or _random
    sf, ef = start_frame - (start_frame % stride), end_frame - (end_frame % stride)
    n_segments = (ef - sf) // stride
    n = 4 * S
    if train:
        sel = rng.choice(n_segments, size=n, replace=(n_segments < n))
    else:
        sel = np.linspace(0, n_segments, num=n, endpoint=False, dtype=int)
    sel = np.sort(sel)
    lo_, hi_ = min(max(sf // stride, 0), video_len), min(max(ef // stride, 0), video_len)
    size = max(hi_ - lo_, 0)
    if size == 0 or sel.max(initial=0) >= size:
        rows = np.full(n, -1, dtype=np.int64)  # np.take raised -> zeros
    else:
        rows = (first_row + lo_ + sel).astype(np.int64)
    rows = rows.reshape(4, S)
    label = int(state_change)
    if train and state_change and py_random.random() < aug_prob:  # (the reference draws the number in exactly this case)
        # the reference's frame-replacement branch adds an ndarray and a list (``graph[:p] + [graph[p-1]] * k``) or hands
        # torch.from_numpy a list: it cannot run as written, and configs/dataset_oscc/ego4d.yaml sets aug_prob: 0
        raise NotImplementedError("oscc_item: the frame-replacement augmentation (aug_prob > 0) is not executable in the reference")
    return {"lo": rows, "y": label, "pos": np.arange(rows.shape[0], dtype=np.int64)}


def pnr_item(first_row: int, video_len: int, start_frame: int, end_frame: int, pnr_frame: int, start_sec: float, end_sec: float,
             stride: int, T: int, train: bool, test: bool = False, rng=np.random):
    """Ego4dPNRDataset.get (ego4d_oscc.py:238-302): T candidate frames over the (randomly cropped, on train) clip,
    each feature interpolated between the two stored rows around it; label = one-hot of the candidate nearest to the
    PNR frame; every node repeats its feature in the 3 segment slots."""
    if train:
        length = rng.uniform(5, 8)
        start_s = start_sec + rng.uniform(8 - length)
        start_frame_ = np.floor(start_s * 30).astype(np.int32)
        end_s = start_s + length
        if end_s > end_sec:
            end_s = end_sec
        end_frame_ = np.floor(end_s * 30).astype(np.int32)
        if pnr_frame > end_frame_:
            end_frame_ = end_frame
        if pnr_frame < start_frame_:
            start_frame_ = start_frame
        start_frame, end_frame = start_frame_, end_frame_
    cand = np.linspace(start_frame, end_frame, num=T, dtype=int, endpoint=False)
    cand = np.clip(cand, start_frame, end_frame)
    lo = np.clip(np.floor(cand / stride).astype(int), 0, video_len - 1)
    hi = np.clip(np.ceil(cand / stride).astype(int), 0, video_len - 1)
    w = (cand % stride) / stride
    if test:
        y = -np.ones(T, dtype=np.int64)
    else:
        y = np.zeros(T, dtype=np.int64)
        y[int(np.abs(cand - pnr_frame).argmin())] = 1
    rep = lambda v: np.repeat(v[:, None], 3, axis=1)
    return {"lo": rep(first_row + lo).astype(np.int64), "hi": rep(first_row + hi).astype(np.int64), "w": rep(w), "y": y,
            "pos": np.arange(T, dtype=np.int64), "start_frame": start_frame, "end_frame": end_frame, "pnr_frame": pnr_frame}


def take_reference(table: np.ndarray, item) -> np.ndarray:
    """numpy evaluation of an item's feature block (what ``FeatureStore.gather`` computes on the device)."""
    lo = item["lo"]
    get = lambda idx: np.where(idx[..., None] >= 0, table[np.maximum(idx, 0)], 0.0).astype(np.float32)
    a = get(lo)
    if "hi" not in item:
        return a
    hi, w = item["hi"], item["w"]
    b = get(hi)
    mix = ((1 - w)[..., None] * a + w[..., None] * b).astype(np.float32)  # double products and sum, rounded once
    return np.where((lo == hi)[..., None], a, mix)
