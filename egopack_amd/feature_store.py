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

    def gather(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """x[..., :] = table[idx[...], :] (zeros where idx < 0): ``[N, S] -> [N, S, F]`` on the device."""
        idx = idx.to(self.table.device, dtype=torch.int64, non_blocking=True).contiguous()
        shape = (*idx.shape, self.features_size)
        if out is None:
            out = torch.empty(shape, dtype=dtype or self.table.dtype, device=self.table.device)
        elif tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError(f"FeatureStore.gather: out must be contiguous {shape}")
        ops._need_gpu(self.table, out)
        n = idx.numel()
        if n:
            ops._ck(_lib.load().egk_gather_rows(ops._stream(), ops._p(self.table), ops._dt(self.table), self.table.stride(0),
                                                self.rows, ops._p(idx), ops._p(out), ops._dt(out), n, self.features_size),
                    "egk_gather_rows")
        return out


def materialise_features(batch, store: FeatureStore, dtype: Optional[torch.dtype] = None):
    """Give a batch that carries ``x_idx`` its ``x`` (no-op when ``x`` is already there)."""
    if getattr(batch, "x", None) is None and getattr(batch, "x_idx", None) is not None:
        batch.x = store.gather(batch.x_idx, dtype=dtype)
    return batch
