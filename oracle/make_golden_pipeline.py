#!/usr/bin/env python3
"""tests/golden/pipeline.pt: the reference datasets' own ``get`` methods (data/ego4d_fho.py:217-242, :353-396,
data/ego4d_oscc.py:191-223, :238-302) run on synthetic videos and annotations (build container only).
TEST INFRASTRUCTURE ONLY.  Usage: python oracle/make_golden_pipeline.py

The methods are called unbound on a stub ``self`` that carries exactly the attributes they read (annotation entries built
from the reference's own namedtuples, a dict of [frames, F] float32 arrays as ``_features``); numpy's and Python's
global random states are seeded and recorded per call.  Stored: the annotations, the random states, and the x / y / pos /
frame outputs.  Data only."""
import importlib.util
import random
import sys
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path = [p for p in sys.path if Path(p or ".").resolve() != REPO]
_spec = importlib.util.spec_from_file_location("oracle", REPO / "oracle" / "__init__.py", submodule_search_locations=[str(REPO / "oracle")])
_oracle = importlib.util.module_from_spec(_spec)
sys.modules["oracle"] = _oracle
_spec.loader.exec_module(_oracle)
sys.path.insert(0, "/root/reference")
from oracle import _pyg_standin  # noqa: E402

_pyg_standin.install()
sys.modules["torch_geometric.data"].Dataset = object

import data.ego4d_fho as fho  # noqa: E402  (reference)
import data.ego4d_oscc as oscc  # noqa: E402  (reference)

assert "/root/reference" in fho.__file__ and "/root/reference" in oscc.__file__
F, STRIDE = 12, 16
rs = np.random.RandomState(0)
videos = {"vidA": rs.standard_normal((90, F)).astype(np.float32), "vidB": rs.standard_normal((40, F)).astype(np.float32)}
out = {"videos": {k: torch.from_numpy(v) for k, v in videos.items()}, "stride": STRIDE, "cases": []}


def dump(d):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(d).items() if v is not None}


def run(kind, cls, stub, idx, entry):
    np.random.seed(1000 + len(out["cases"]))
    random.seed(2000 + len(out["cases"]))
    state = (np.random.get_state(), random.getstate())
    d = cls.get(stub, idx)
    out["cases"].append({"kind": kind, "entry": entry, "np_state": state[0], "py_state": state[1], "split": stub.split, "data": dump(d)})


# ---- AR: windows of 5 actions, incl. an empty and a past-the-end window ------------------------------------------------
acts = [fho.Ego4dFHOEntry(i, "vidA", "clipA", s, e, v, n) for i, (s, e, v, n) in enumerate(
    [(0, 160, 3, 5), (160, 400, 1, 2), (400, 400, 0, 7), (500, 1100, 6, 1), (1200, 1500, 2, 9), (1380, 2000, 4, 4)])]
for split, rnd in (("train", True), ("train", False), ("val", True)):
    for centre in (2, 3):
        window = [acts[max(0, min(len(acts) - 1, centre + o))] for o in (-2, -1, 0, 1, 2)]
        stub = types.SimpleNamespace(action_segments=[fho.Ego4dAREntry("vidA", "clipA", window)], window_size=5, _features=videos,
                                     stride=STRIDE, split=split, randomize_train=rnd, num_segments=3, features_size=F)
        run("ar", fho.Ego4dRecognitionDataset, stub, 0,
            {"video": "vidA", "window_size": 5, "randomize_train": rnd, "actions": [(a.start_frame, a.end_frame, a.verb_label, a.noun_label) for a in window]})
# ---- LTA: 2 input clips + 4 forecast nodes ('avg' and 'zero') -----------------------------------------------------------
for split in ("train", "val"):
    for mode in ("avg", "zero"):
        entry = fho.Ego4dLTAEntry("vidB", "clipB", 1, acts[0:2], acts[2:6])
        stub = types.SimpleNamespace(lta_annotations=[entry], n_forecast_clips=4, n_input_clips=2, _features=videos, stride=STRIDE,
                                     split=split, num_segments=3, features_size=F, append_node=mode)
        run("lta", fho.Ego4dLTADataset, stub, 0,
            {"video": "vidB", "append_node": mode, "input": [(a.start_frame, a.end_frame) for a in entry.input_clips],
             "forecast_labels": [(a.verb_label, a.noun_label) for a in entry.forecast_clips]})
# ---- OSCC (aug_prob 0 as configured) and PNR -----------------------------------------------------------------------------
fields = oscc.Ego4dOSCCPNREntry._fields
print("OSCC entry fields:", fields)


def seg(**kw):
    base = dict.fromkeys(fields)
    base.update(kw)
    return oscc.Ego4dOSCCPNREntry(**base)


segs = [seg(video_uid="vidA", unique_uid="u0", start_frame=35, end_frame=275, pnr_frame=170, state_change=1, start_sec=35 / 30, end_sec=275 / 30),
        seg(video_uid="vidA", unique_uid="u1", start_frame=600, end_frame=840, pnr_frame=700, state_change=0, start_sec=20.0, end_sec=28.0),
        seg(video_uid="vidB", unique_uid="u2", start_frame=300, end_frame=700, pnr_frame=420, state_change=1, start_sec=10.0, end_sec=23.3)]
for split in ("train", "validation"):
    for i, sg in enumerate(segs):
        stub = types.SimpleNamespace(annotations=segs, _features=videos, stride=STRIDE, split=split, num_segments=3, aug_prob=0.0)
        run("oscc", oscc.Ego4dOSCCDataset, stub, i, sg._asdict())
        stub = types.SimpleNamespace(annotations=segs, _features=videos, stride=STRIDE, split=split, num_segments=16)
        run("pnr", oscc.Ego4dPNRDataset, stub, i, sg._asdict())
torch.save(out, REPO / "tests" / "golden" / "pipeline.pt")
print("pipeline.pt:", len(out["cases"]), "cases:", {k: sum(c["kind"] == k for c in out["cases"]) for k in ("ar", "lta", "oscc", "pnr")})
