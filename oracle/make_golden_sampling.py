#!/usr/bin/env python3
"""tests/golden/sampling.pt: the reference's segment sampling (data/base_dataset.py:128-155 BaseFrameDataset) run as
it is on seeded numpy state (build container only).  TEST INFRASTRUCTURE ONLY.  Usage: python oracle/make_golden_sampling.py"""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path = [p for p in sys.path if Path(p or ".").resolve() != REPO]
sys.path.insert(0, "/root/reference")
from data.base_dataset import BaseFrameDataset  # noqa: E402  (reference)

assert "/root/reference" in sys.modules["data.base_dataset"].__file__
cases = []
np.random.seed(1234)
for size, n in [(0, 3), (1, 3), (2, 3), (3, 3), (4, 3), (7, 3), (8, 8), (17, 8), (100, 3), (5, 1), (9, 4), (64, 3)]:
    state = np.random.get_state()
    rnd = BaseFrameDataset.random_sampling_indices(size, n)
    uni = BaseFrameDataset.uniform_sampling_indices(size, n)
    cases.append({"size": size, "n": n, "state_before": state, "random": rnd.tolist(), "uniform": uni.tolist(),
                  "state_after_pos": int(np.random.get_state()[2])})
# a full window take on a small feature array, incl. the failure -> zero-clip path of ego4d_fho.py:228-238
feats = np.arange(40 * 5, dtype=np.float32).reshape(40, 5)
takes = []
np.random.seed(99)
for (a, b, n, rnd) in [(3, 20, 3, True), (3, 20, 3, False), (10, 10, 3, True), (38, 45, 3, False), (0, 2, 3, True), (39, 39, 4, False),
                       (5, 6, 3, True), (12, 30, 8, True)]:
    state = np.random.get_state()
    try:
        out = (BaseFrameDataset.random_sampling if rnd else BaseFrameDataset.uniform_sampling)(feats[a:b], n)
    except Exception:  # noqa: BLE001  (the reference catches everything and substitutes zeros)
        out = np.zeros((n, feats.shape[1]), dtype=np.float32)
    takes.append({"a": a, "b": b, "n": n, "random": rnd, "state_before": state, "out": torch.from_numpy(np.array(out))})
torch.save({"index_cases": cases, "features": torch.from_numpy(feats), "takes": takes}, REPO / "tests" / "golden" / "sampling.pt")
print("sampling.pt:", len(cases), "index cases,", len(takes), "window takes")
