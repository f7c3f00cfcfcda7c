#!/usr/bin/env python3
"""Generate tests/golden/meters.pt and tests/golden/validate.pt by running the REFERENCE's own meter / validation
code (build container only).  TEST INFRASTRUCTURE ONLY.  Usage: python oracle/make_golden_meters.py

Executed from /root/reference, unmodified:
  * utils/meters/utils.py ``topk_accuracy`` / ``topk_recall`` (numpy);
  * utils/meters/ego4d.py ``Ego4dPNRMeter.update`` (localisation error) and ``Ego4dLTAMeter.update`` (the 22 x 5
    reshape, the 2 dropped nodes, min over K) -- torchmetrics classes are inert stand-ins (absent package) and
    ``editdistance.eval`` is the oracle's Levenshtein (absent package; what the fixture pins is the reference's
    bookkeeping around it);
  * validate.py ``validate`` / ``validate_lta`` / ``validate_pnr`` with the reference models and a recording meter:
    what reaches ``meter.update`` per batch (logits, labels, loss, predictions shapes) is the fixture.
Only data is stored.
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path = [p for p in sys.path if Path(p or ".").resolve() != REPO]
_spec = importlib.util.spec_from_file_location("oracle", REPO / "oracle" / "__init__.py",
                                               submodule_search_locations=[str(REPO / "oracle")])
_oracle = importlib.util.module_from_spec(_spec)
sys.modules["oracle"] = _oracle
_spec.loader.exec_module(_oracle)
sys.path.insert(0, str(REF))

import torch  # noqa: E402

from oracle import _pyg_standin, meters as OM, pyg_ops as P  # noqa: E402

_pyg_standin.install()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Inert:  # torchmetrics metric stand-in: accepts everything, records nothing
    def __init__(self, *a, **k):
        self.values = []

    def to(self, *_a, **_k):
        return self

    def update(self, *a, **k):
        self.values.append(a[0].detach().clone() if a and torch.is_tensor(a[0]) else a)

    __call__ = update

    def compute(self):
        return torch.tensor(0.0)


_mod("torchmetrics.aggregation", MeanMetric=_Inert)
_mod("torchmetrics.classification", MulticlassAccuracy=_Inert, ConfusionMatrix=_Inert, MulticlassCalibrationError=_Inert,
     BinaryRecall=_Inert, BinaryAccuracy=_Inert, BinaryAUROC=_Inert)
_mod("torchmetrics.functional.classification", multiclass_recall=None)
_mod("torchmetrics.functional", classification=sys.modules["torchmetrics.functional.classification"])
_mod("torchmetrics", aggregation=sys.modules["torchmetrics.aggregation"], classification=sys.modules["torchmetrics.classification"],
     functional=sys.modules["torchmetrics.functional"], MeanMetric=_Inert, CatMetric=_Inert, SumMetric=_Inert, Metric=object)
sys.modules["wandb"].Table = object
sys.modules["wandb"].plot = None
_mod("sklearn.manifold", TSNE=None)
_mod("editdistance", eval=lambda a, b: OM.levenshtein(list(a), list(b)))
_mod("utils.confusion", Top2ConfusionMatrix=_Inert)
for name, classes in (("data.ego4d_fho", ("Ego4dRecognitionDataset", "Ego4dAnticipationDataset", "Ego4dLTADataset")),
                      ("data.ego4d_oscc", ("Ego4dOSCCDataset", "Ego4dPNRDataset"))):
    _mod(name, **{c: type(c, (), {}) for c in classes})


def _unbatch(src, batch):
    sizes = torch.bincount(batch).tolist()
    return src.split(sizes)


sys.modules["torch_geometric.utils"].unbatch = _unbatch
sys.modules["torch_geometric.loader.dataloader"].DataLoader = object

import utils.meters.utils as ref_utils  # noqa: E402  (reference)
import utils.meters.ego4d as ref_meters  # noqa: E402  (reference)
import validate as ref_validate  # noqa: E402  (reference)
from models.graph import Graph  # noqa: E402
from models.tasks import RecognitionTask, LTATask, PNRTask, OSCCTask  # noqa: E402

assert str(REF) in ref_utils.__file__ and str(REF) in ref_meters.__file__ and str(REF) in ref_validate.__file__
OUT = REPO / "tests" / "golden"


class _DS:
    label_names = ["verbs", "nouns"]
    class_labels = [[f"v{i}" for i in range(7)], [f"n{i}" for i in range(11)]]


def golden_meters():
    g = torch.Generator().manual_seed(7)
    out = {}
    # ---- top-k accuracy / recall (reference numpy functions) -----------------------------------------------------
    scores = torch.randn(96, 11, generator=g).numpy()
    labels = torch.randint(0, 8, (96,), generator=g).numpy()  # classes 8..10 never occur
    out["topk"] = {"scores": torch.from_numpy(scores), "labels": torch.from_numpy(labels), "ks": (1, 2, 3, 5),
                   "accuracy": [float(v) for v in ref_utils.topk_accuracy(scores, labels, (1, 2, 3, 5))],
                   "recall": {k: float(ref_utils.topk_recall(scores, labels, k=k)) for k in (1, 2, 5)},
                   "class3": [float(v) for v in ref_utils.topk_accuracy(scores, labels, (1, 5), selected_class=3)]}
    # ---- PNR localisation error (reference meter class) ------------------------------------------------------------
    B, T = 6, 16
    logits = torch.randn(B * T, generator=g)
    batch = torch.arange(B).repeat_interleave(T)
    sf = torch.randint(0, 1000, (B,), generator=g)
    ef = sf + torch.randint(200, 300, (B,), generator=g)
    pf = sf + torch.randint(0, 240, (B,), generator=g)
    y = torch.zeros(B * T, dtype=torch.long)
    m = ref_meters.Ego4dPNRMeter(object(), device=torch.device("cpu"))
    m.update(logits, y, batch, sf, ef, pf, torch.tensor(0.5))
    out["pnr"] = {"logits": logits, "batch": batch, "start_frame": sf, "end_frame": ef, "pnr_frame": pf,
                  "loc_errors": torch.tensor(m.loc_errors, dtype=torch.float64)}
    # ---- LTA edit distance bookkeeping (reference meter class; Levenshtein itself is the oracle's) -----------------
    n_seq = 5
    labels = torch.randint(0, 7, (n_seq * 22, 2), generator=g)
    labels[:, 1] = torch.randint(0, 11, (n_seq * 22,), generator=g)
    preds = [torch.randint(0, 7, (n_seq * 22, 5), generator=g), torch.randint(0, 11, (n_seq * 22, 5), generator=g)]
    preds[0][:22, 0] = labels[:22, 0]  # one exact sample -> distance 0 for sequence 0
    lm = ref_meters.Ego4dLTAMeter(_DS(), device=torch.device("cpu"))
    lm.update((torch.randn(n_seq * 22, 7, generator=g), torch.randn(n_seq * 22, 11, generator=g)), labels, preds,
              torch.tensor(0.5))
    out["lta"] = {"labels": labels, "predictions": preds,
                  "verbs": lm.verbs_edit_distance.values[0].double(), "nouns": lm.nouns_edit_distance.values[0].double()}
    torch.save(out, OUT / "meters.pt")
    print("meters.pt:", {k: list(v.keys()) for k, v in out.items()})


class _Recorder:
    def __init__(self):
        self.calls = []

    def update(self, *args):
        self.calls.append([a.detach().clone() if torch.is_tensor(a) else
                           ([t.detach().clone() for t in a] if isinstance(a, (tuple, list)) else a) for a in args])


def golden_validate():
    """Reference validate loops on tiny reference models: what reaches meter.update."""
    from oracle import pyg_ops
    g = torch.Generator().manual_seed(11)
    F_IN, S, H, HP, HEADS = 48, 3, 32, 40, (7, 11)
    trn = {"_target_": "models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": HP}
    torch.manual_seed(3)
    model = Graph(F_IN, hidden_size=H, depth=2, temporal_pooling=trn, num_segments=S)
    ar = RecognitionTask(H, H, HEADS)
    lta = LTATask(H, H, HEADS)
    pnr = PNRTask(H, H)
    oscc = OSCCTask(H, H)

    def sd(m):
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    def batch_of(B, T, kind):
        x = torch.randn(B * T, S, F_IN, generator=g)
        pos = torch.arange(T).repeat(B)
        bvec = torch.arange(B).repeat_interleave(T)
        ei = torch.cat([pyg_ops.radius_graph(torch.arange(T).float(), 1.5) + T * b for b in range(B)], 1)
        d = pyg_ops.OData(x=x, pos=pos, edge_index=ei, batch=bvec, num_graphs=B)
        if kind in ("ar", "lta"):
            d.y = torch.stack([torch.randint(0, HEADS[0], (B * T,), generator=g), torch.randint(0, HEADS[1], (B * T,), generator=g)], 1)
            if kind == "ar":
                d.y[torch.rand(B * T, generator=g) < 0.5] = -1
        elif kind == "oscc":
            d.y = torch.randint(0, 2, (B,), generator=g)
        else:
            d.y = torch.zeros(B * T, dtype=torch.long)
            d.y[torch.arange(B) * T + torch.randint(0, T, (B,), generator=g)] = 1
            d.start_frame = torch.randint(0, 100, (B,), generator=g)
            d.end_frame = d.start_frame + 240
            d.pnr_frame = d.start_frame + torch.randint(0, 240, (B,), generator=g)
        d.to = lambda *_a, **_k: d
        return d

    def dump(d):
        return {k: v.clone() for k, v in vars(d).items() if torch.is_tensor(v)} | {"num_graphs": d.num_graphs}

    out = {"dims": (F_IN, S, H, HP, HEADS), "sd": {"model": sd(model), "ar": sd(ar), "lta": sd(lta), "pnr": sd(pnr), "oscc": sd(oscc)}}
    for name, fn, task, kind in (("ar", ref_validate.validate, ar, "ar"), ("oscc", ref_validate.validate, oscc, "oscc"),
                                 ("pnr", ref_validate.validate_pnr, pnr, "pnr")):
        batches = [batch_of(3, 6, kind), batch_of(2, 6, kind)]
        rec = _Recorder()
        if fn is ref_validate.validate:
            fn(0, model, batches, rec, task, device="cpu")
        else:
            fn(model, batches, rec, task, device="cpu")
        out[name] = {"batches": [dump(b) for b in batches], "calls": rec.calls}
    # LTA: sampling uses the global torch RNG -> fix it right before the call
    batches = [batch_of(2, 22, "lta")]
    rec = _Recorder()
    torch.manual_seed(99)
    ref_validate.validate_lta(model, batches, rec, lta, device="cpu")
    out["lta"] = {"batches": [dump(b) for b in batches], "calls": rec.calls, "seed": 99}
    torch.save(out, OUT / "validate.pt")
    print("validate.pt:", [(k, len(v["calls"])) for k, v in out.items() if isinstance(v, dict) and "calls" in v])


if __name__ == "__main__":
    golden_meters()
    golden_validate()
