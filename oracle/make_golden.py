#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REFERENCE's own Python (build container only).

TEST INFRASTRUCTURE ONLY.  Usage (from anywhere):  python oracle/make_golden.py

/root/reference is imported read-only with the stand-ins of oracle/_pyg_standin.py registered
for the absent third-party packages (torch_geometric 2.3.0, hydra, wandb, torchmetrics,
torchvision); the reference's own modules and train loops then run unmodified on small seeded
inputs (F=48, H=32, Hp=40, S=3, heads (7, 11), K=37 prototypes, kg=4) and their inputs,
parameters, outputs and gradients are stored as fixtures (each well under 1 MB).  The fixtures
are data only: no reference source text is stored.  /root/reference never travels to the GPU
box; the fixtures and this script do.
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")

# The repo has same-named top-level packages (models/, criterion/, utils/, graphone.py): keep the
# repo root OFF sys.path and load ``oracle`` by file location, then put the reference first.
sys.path = [p for p in sys.path if Path(p or ".").resolve() != REPO]
_spec = importlib.util.spec_from_file_location(
    "oracle", REPO / "oracle" / "__init__.py", submodule_search_locations=[str(REPO / "oracle")])
_oracle = importlib.util.module_from_spec(_spec)
sys.modules["oracle"] = _oracle
_spec.loader.exec_module(_oracle)
sys.path.insert(0, str(REF))

import torch  # noqa: E402

from oracle import _pyg_standin, pyg_ops as P  # noqa: E402

_pyg_standin.install()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Meter:  # stand-in for torchmetrics.aggregation.MeanMetric (reporting only, off the path)
    def __init__(self, *a, **k):
        self.values = []

    def to(self, *_a, **_k):
        return self

    def update(self, v):
        self.values.append(v.detach().clone())


# reporting / evaluation modules of the reference that are off the hot path and need absent
# packages (torchmetrics, editdistance, pandas readers of the licensed dataset)
_mod("torchmetrics.aggregation", MeanMetric=_Meter)
_mod("torchmetrics", aggregation=sys.modules["torchmetrics.aggregation"])
_mod("utils.meters", build_meter_for_dataset=None)
_mod("validate", validate=None, validate_lta=None, validate_pnr=None)
_mod("utils.wandb", format_wandb_run_name=None)
_mod("torch_geometric.transforms.radius_graph", RadiusGraph=None)

from models.graph import Graph  # noqa: E402  (reference)
from models.temporal_pooling.trn_pooling import TRNPooling  # noqa: E402
from models.tasks import RecognitionTask, OSCCTask, LTATask, PNRTask  # noqa: E402
from models.graphONE.graphONE import GraphONE, cos_dissimilarity  # noqa: E402
from models.transforms.lta_temp_connectivity import LTATemporalConnectivity  # noqa: E402
from criterion.wrapper import MetricSelectorWrapper  # noqa: E402
from graphone import build_graphone  # noqa: E402
from utils.dataloading import multiloader  # noqa: E402
import main_temporal  # noqa: E402
import main_egopack  # noqa: E402

assert Graph.__module__ == "models.graph" and str(REF) in sys.modules["models.graph"].__file__

OUT = REPO / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)

F_IN, S, H, HP, HEADS, K_PROTO, KG = 48, 3, 32, 40, (7, 11), 37, 4
TRN_CFG = {"_target_": "models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": HP}


def sd_of(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def grads_of(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def randomize_norm_affine(m, g):
    """LayerNorm affine params init to (1, 0); perturb so that parity tests exercise them."""
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() == 1 and ("bias" in name or "weight" in name):
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def make_sample(kind, T, g, verb_zero_at=None):
    """One un-batched sample shaped like the reference datasets emit."""
    x = torch.randn(T, S, F_IN, generator=g)
    if kind == "ar":  # centre node labelled (ego4d_fho.py:222-224)
        pos = torch.arange(T) - T // 2
        y = torch.full((T, 2), -1, dtype=torch.long)
        y[T // 2, 0] = torch.randint(0, HEADS[0], (1,), generator=g)
        y[T // 2, 1] = torch.randint(0, HEADS[1], (1,), generator=g)
    elif kind == "lta":  # first 2 nodes unlabelled inputs, rest forecast (ego4d_fho.py:361-362)
        pos = torch.arange(T)
        y = torch.stack([torch.randint(1, HEADS[0], (T,), generator=g),
                         torch.randint(0, HEADS[1], (T,), generator=g)], 1)
        y[:2] = -1
        if verb_zero_at is not None:
            y[verb_zero_at, 0] = 0
    elif kind == "oscc":
        pos = torch.arange(T)
        y = int(torch.randint(0, 2, (1,), generator=g))
    elif kind == "pnr":
        pos = torch.arange(T)
        y = torch.zeros(T, dtype=torch.long)
        y[int(torch.randint(0, T, (1,), generator=g))] = 1
    else:
        raise ValueError(kind)
    return _pyg_standin.Data(x=x, pos=pos, y=y, batch=None)


def make_batch(kind, B, T, g, k=1, verb_zero=False):
    samples = []
    lta_tf = LTATemporalConnectivity(r=k + 0.5, loop=False)  # reference transform
    for b in range(B):
        s = make_sample(kind, T, g, verb_zero_at=(4 if (verb_zero and b == 0) else None))
        if kind == "lta":
            s = lta_tf(s)
        else:  # RadiusGraph(r=k+0.5, loop=False): PyG transform, restated in oracle/pyg_ops.py
            s.edge_index = P.radius_graph(s.pos, k + 0.5, None, False, 32)
        samples.append(s)
    batch = P.collate(samples)
    return _pyg_standin.Data(**batch.__dict__)


def batch_dict(d):
    return {k: v.clone() if torch.is_tensor(v) else v for k, v in d.__dict__.items()}


def save(name, obj):
    path = OUT / f"{name}.pt"
    torch.save(obj, path)
    print(f"wrote {path.relative_to(REPO)}  ({path.stat().st_size / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------
def golden_trn():
    g = torch.Generator().manual_seed(101)
    torch.manual_seed(101)
    m = TRNPooling(F_IN, H, S, hidden_size=HP, dropout=0.0)
    randomize_norm_affine(m, g)
    x = torch.randn(10, S, F_IN, generator=g, requires_grad=True)
    w = torch.randn(10, H, generator=g)
    out = m(x)
    (out * w).sum().backward()
    save("trn_pooling", {"sd": sd_of(m), "x": x.detach(), "w": w, "out": out.detach(),
                         "grad_x": x.grad.clone(), "grads": grads_of(m)})


def golden_graph():
    g = torch.Generator().manual_seed(202)
    torch.manual_seed(202)
    m = Graph(F_IN, hidden_size=H, depth=3, pre_dropout=0, temporal_pooling=TRN_CFG, num_segments=S)
    randomize_norm_affine(m, g)
    cases = {}
    for name, kind, B, T, k, vz in [("ar_T9_k1", "ar", 2, 9, 1, False), ("lta_T22_k1", "lta", 2, 22, 1, True),
                                    ("oscc_T4_k2", "oscc", 3, 4, 2, False), ("pnr_T16_k2", "pnr", 2, 16, 2, False)]:
        d = make_batch(kind, B, T, g, k, vz)
        m.zero_grad()
        out = m(d)
        w = torch.randn(out.shape, generator=g)
        (out * w).sum().backward()
        cases[name] = {"data": batch_dict(d), "w": w, "out": out.detach(), "grads": grads_of(m)}
    save("graph_forward", {"sd": sd_of(m), "depth": 3, "cases": cases})


def golden_heads():
    g = torch.Generator().manual_seed(303)
    torch.manual_seed(303)
    N, B = 12, 3
    feat = torch.randn(N, H, generator=g)
    batch = torch.arange(B).repeat_interleave(N // B)
    aux = {t: torch.randn(N, H, generator=g) for t in ("lta", "oscc", "pnr")}
    y2 = torch.stack([torch.randint(0, HEADS[0], (N,), generator=g), torch.randint(0, HEADS[1], (N,), generator=g)], 1)
    y2[::3] = -1
    out = {"feat": feat, "batch": batch, "y2": y2}

    for cls, key in ((RecognitionTask, "ar"), (LTATask, "lta")):
        for avg in (False, True):
            others = tuple(t for t in ("ar", "lta", "oscc", "pnr") if t != key)
            aux_k = {t: torch.randn(N, H, generator=g) for t in others}
            t = cls(H, H, HEADS, dropout=0, head_dropout=0, aux_tasks=others, average_logits=avg)
            randomize_norm_affine(t, g)
            f = t.forward_features(feat)
            plain = t.forward_logits(f)
            fused = t.forward_logits(f, None, aux_k)
            loss = t.compute_loss(fused, y2)
            out[f"{key}_avg{int(avg)}"] = {"sd": sd_of(t), "aux": aux_k, "features": f.detach(),
                                           "logits": [l.detach() for l in plain],
                                           "logits_fused": [l.detach() for l in fused], "loss": loss.detach()}

    class _DS:  # what MetricSelectorWrapper needs of a dataset (criterion/wrapper.py:36,69-80)
        has_joint_label = False
        num_labels = 2
    crit = MetricSelectorWrapper(torch.nn.CrossEntropyLoss(reduction="none", ignore_index=-1), _DS())
    logits = (torch.randn(N, HEADS[0], generator=g), torch.randn(N, HEADS[1], generator=g))
    out["selector"] = {"logits": list(logits), "loss": crit(logits, y2)}

    yb = torch.randint(0, 2, (B,), generator=g)
    for kind in ("ce", "bce"):
        for avg in (False, True):
            t = OSCCTask(H, H, 0, 0, loss_func=kind, aux_tasks=("ar", "lta", "pnr"), average_logits=avg)
            randomize_norm_affine(t, g)
            aux_k = {k: torch.randn(N, H, generator=g) for k in ("ar", "lta", "pnr")}
            f = t.forward_features(feat)
            plain = t.forward_logits(f, batch)
            fused = t.forward_logits(f, batch, aux_k)
            out[f"oscc_{kind}_avg{int(avg)}"] = {"sd": sd_of(t), "aux": aux_k, "y": yb, "logits": plain.detach(),
                                                 "logits_fused": fused.detach(),
                                                 "loss": t.compute_loss(fused, yb).detach()}
    yp = torch.zeros(N, dtype=torch.long)
    yp[[2, 7]] = 1
    for avg in (False, True):
        t = PNRTask(H, H, 0, 0, aux_tasks=("ar", "oscc", "lta"), average_logits=avg)
        randomize_norm_affine(t, g)
        aux_k = {k: torch.randn(N, H, generator=g) for k in ("ar", "oscc", "lta")}
        f = t.forward_features(feat)
        plain = t.forward_logits(f)
        fused = t.forward_logits(f, aux_k)
        out[f"pnr_avg{int(avg)}"] = {"sd": sd_of(t), "aux": aux_k, "y": yp, "logits": plain.detach(),
                                     "logits_fused": fused.detach(), "loss": t.compute_loss(fused, yp).detach()}
    save("heads", out)


def golden_graphone():
    g = torch.Generator().manual_seed(404)
    torch.manual_seed(404)
    banks = {t: torch.randn(K_PROTO, H, generator=g) for t in ("ar", "lta", "pnr")}
    N = 14
    out = {"banks": banks, "k": KG}
    d = cos_dissimilarity(torch.randn(5, H, generator=g), banks["ar"])
    out["cos_example"] = d
    for residual in (False, True):
        m = GraphONE({k: v.clone() for k, v in banks.items()}, features_size=H, hidden_size=H, k=KG, depth=2,
                     residual=residual, dropout=0, output_dropout=0, output_projection=True)
        randomize_norm_affine(m, g)
        feats = {t: torch.randn(N, H, generator=g, requires_grad=True) for t in ("ar", "lta", "pnr")}
        res, closest = m.interact(feats)
        w = {t: torch.randn(N, H, generator=g) for t in feats}
        sum((res[t] * w[t]).sum() for t in feats).backward()
        out[f"residual{int(residual)}"] = {
            "sd": sd_of(m), "depth": 2, "features": {t: f.detach() for t, f in feats.items()}, "w": w,
            "out": {t: r.detach() for t, r in res.items()},
            "closest": {t: [c.clone() for c in cs] for t, cs in closest.items()},
            "grad_features": {t: f.grad.clone() for t, f in feats.items()}, "grads": grads_of(m)}
    save("graphone", out)


def golden_build_graphone():
    g = torch.Generator().manual_seed(505)
    torch.manual_seed(505)
    model = Graph(F_IN, hidden_size=H, depth=3, pre_dropout=0, temporal_pooling=TRN_CFG, num_segments=S)
    ar = RecognitionTask(H, H, HEADS)
    lta = LTATask(H, H, HEADS)
    pnr = PNRTask(H, H)
    for m in (model, ar, lta, pnr):
        randomize_norm_affine(m, g)
    batches = [make_batch("ar", 6, 9, g, 1) for _ in range(3)]
    # repeat some labels so that banks average several rows
    batches[1].y[batches[1].y[:, 0] != -1] = batches[0].y[batches[0].y[:, 0] != -1]
    banks = build_graphone(model, ar, [ar, lta, pnr], batches, device="cpu")
    save("build_graphone", {"backbone": sd_of(model), "tasks": {"ar": sd_of(ar), "lta": sd_of(lta), "pnr": sd_of(pnr)},
                            "batches": [batch_dict(b) for b in batches], "n_classes": HEADS,
                            "banks": {k: v.clone() for k, v in banks.items()}})


def golden_edges_and_loader():
    g = torch.Generator().manual_seed(606)
    out = {}
    tf = LTATemporalConnectivity(r=1.5, loop=False)
    for name, T, vz in (("lta_T22", 22, None), ("lta_T22_verb0", 22, 5), ("lta_T8_verb0_first", 8, 2)):
        s = make_sample("lta", T, g, verb_zero_at=vz)
        s = tf(s)
        out[name] = {"pos": s.pos, "y": s.y, "r": 1.5, "edge_index": s.edge_index}
    tf2 = LTATemporalConnectivity(r=2.5, loop=False)
    s = tf2(make_sample("lta", 12, g))
    out["lta_T12_r2.5"] = {"pos": s.pos, "y": s.y, "r": 2.5, "edge_index": s.edge_index}
    # multiloader restart semantics (utils/dataloading.py:8-47): loaders of different lengths
    seq = [tuple(x) for x in multiloader([[1, 2, 3], [10, 20], None, [7]], [1.0, 1.0, 1.0, 1.0])]
    seq2 = [tuple(x) for x in multiloader([[1, 2], [10, 20, 30], [5], [7]], [1.0, 0.0, 1.0, 1.0])]
    out["multiloader"] = {"case1": seq, "case2": seq2}
    save("edges_loader", out)


def _mtl_modules(g, aux=False):
    model = Graph(F_IN, hidden_size=H, depth=3, pre_dropout=0, temporal_pooling=TRN_CFG, num_segments=S)
    if aux:
        ar = RecognitionTask(H, H, HEADS, aux_tasks=("oscc", "lta", "pnr"))
        oscc = OSCCTask(H, H, aux_tasks=("ar", "lta", "pnr"), average_logits=True)
        lta = LTATask(H, H, HEADS, aux_tasks=("ar", "oscc", "pnr"))
        pnr = PNRTask(H, H, aux_tasks=("ar", "oscc", "lta"))
    else:
        ar, oscc, lta, pnr = RecognitionTask(H, H, HEADS), OSCCTask(H, H), LTATask(H, H, HEADS), PNRTask(H, H)
    for m in (model, ar, oscc, lta, pnr):
        randomize_norm_affine(m, g)
    return model, ar, oscc, lta, pnr


def golden_mtl_train():
    """Runs the reference main_temporal.train (main_temporal.py:49-134) for one 'epoch' of 2
    iterations with AR+LTA+PNR enabled (OSCC weight 0, as in BASELINE config 3) and real
    torch.optim.Adam; stores parameters before/after and the per-iteration loss vectors."""
    g = torch.Generator().manual_seed(707)
    torch.manual_seed(707)
    model, ar, oscc, lta, pnr = _mtl_modules(g)

    class _DS:
        has_joint_label = False
        num_labels = 2
    ce = torch.nn.CrossEntropyLoss(reduction="none", ignore_index=-1)
    crit_ar, crit_lta = MetricSelectorWrapper(ce, _DS()), MetricSelectorWrapper(ce, _DS())
    crit_oscc = torch.nn.CrossEntropyLoss(reduction="none", ignore_index=-1)
    crit_pnr = torch.nn.BCEWithLogitsLoss(reduction="none")
    dl_ar = [make_batch("ar", 3, 9, g, 1) for _ in range(2)]
    dl_lta = [make_batch("lta", 2, 22, g, 1, verb_zero=True) for _ in range(2)]
    dl_oscc = [make_batch("oscc", 3, 4, g, 1) for _ in range(2)]
    dl_pnr = [make_batch("pnr", 2, 16, g, 1) for _ in range(2)]
    before = {"temporal_graph": sd_of(model), "task/recognition": sd_of(ar), "task/oscc": sd_of(oscc),
              "task/lta": sd_of(lta), "task/pnr": sd_of(pnr)}
    params = [*model.parameters(), *ar.parameters(), *oscc.parameters(), *lta.parameters(), *pnr.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5)
    meters = [_Meter() for _ in range(4)]
    main_temporal.train(1, model, opt, ar, dl_ar, crit_ar, meters[0], oscc, dl_oscc, crit_oscc, meters[1],
                        lta, dl_lta, crit_lta, meters[2], pnr, dl_pnr, crit_pnr, meters[3],
                        weight_ar=1.0, weight_oscc=0, weight_lta=0.5, weight_pnr=2.0, device="cpu")
    after = {"temporal_graph": sd_of(model), "task/recognition": sd_of(ar), "task/oscc": sd_of(oscc),
             "task/lta": sd_of(lta), "task/pnr": sd_of(pnr)}
    save("mtl_train", {"before": before, "after": after, "lr": 1e-3, "weight_decay": 1e-5,
                       "weights": {"ar": 1.0, "oscc": 0.0, "lta": 0.5, "pnr": 2.0},
                       "batches": {"ar": [batch_dict(b) for b in dl_ar], "lta": [batch_dict(b) for b in dl_lta],
                                   "oscc": [batch_dict(b) for b in dl_oscc], "pnr": [batch_dict(b) for b in dl_pnr]},
                       "loss_vectors": {"ar": meters[0].values, "oscc": meters[1].values,
                                        "lta": meters[2].values, "pnr": meters[3].values}})


def golden_egopack_train():
    """Runs the reference main_egopack.train (main_egopack.py:64-159) for 2 iterations with OSCC
    as the novel task (BASELINE config 4): frozen AR/LTA/PNR banks, GraphONE k=4 depth=2 residual,
    backbone in eval mode with gradients enabled (defaults.yaml:62-64)."""
    g = torch.Generator().manual_seed(808)
    torch.manual_seed(808)
    model, ar, oscc, lta, pnr = _mtl_modules(g, aux=True)
    banks = {t: torch.randn(K_PROTO, H, generator=g) for t in ("ar", "lta", "pnr")}
    gone = GraphONE({k: v.clone() for k, v in banks.items()}, features_size=H, hidden_size=H, k=KG, depth=2,
                    residual=True, dropout=0, output_dropout=0, output_projection=True,
                    distance_func="cosine", update_edges_interval=1, share_params=False)
    randomize_norm_affine(gone, g)
    dl_oscc = [make_batch("oscc", 3, 4, g, 1) for _ in range(2)]
    before = {"temporal_graph": sd_of(model), "task/recognition": sd_of(ar), "task/oscc": sd_of(oscc),
              "task/lta": sd_of(lta), "task/pnr": sd_of(pnr), "graphone": sd_of(gone)}
    params = [*model.parameters(), *ar.parameters(), *oscc.parameters(), *lta.parameters(), *pnr.parameters(),
              *gone.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5)
    meters = [_Meter() for _ in range(4)]
    main_egopack.tqdm = lambda it: it
    main_egopack.train(1, model, gone, True, opt, ar, None, meters[0], oscc, dl_oscc, meters[1],
                       lta, None, meters[2], pnr, None, meters[3],
                       weight_ar=0, weight_oscc=1.0, weight_lta=0, weight_pnr=0,
                       backprop_temporal_graph=True, temporal_graph_train_mode=False, device="cpu")
    after = {"temporal_graph": sd_of(model), "task/recognition": sd_of(ar), "task/oscc": sd_of(oscc),
             "task/lta": sd_of(lta), "task/pnr": sd_of(pnr), "graphone": sd_of(gone)}
    save("egopack_train", {"before": before, "after": after, "lr": 1e-3, "weight_decay": 1e-5, "k": KG, "depth": 2,
                           "residual": True, "batches": {"oscc": [batch_dict(b) for b in dl_oscc]},
                           "loss_vectors": {"oscc": meters[1].values}})


def golden_variants():
    """Variants of the cited functions beside the configured ones: GraphONE distance_func='l2' (graphONE.py:126-127,
    144-145) and trainable prototypes freeze=False (:47-49), OSCC loss_func='bce' with its gradients (oscc.py:91-93).
    ('focal' needs torchvision.ops.sigmoid_focal_loss: absent package, restated in oracle/path.py, parity unpinned.)"""
    g = torch.Generator().manual_seed(909)
    torch.manual_seed(909)
    banks = {t: torch.randn(K_PROTO, H, generator=g) for t in ("ar", "lta", "pnr")}
    N = 14
    out = {"banks": banks, "k": KG, "depth": 2}
    for name, kw in (("l2", dict(distance_func="l2", freeze=True)), ("trainable", dict(distance_func="cosine", freeze=False)),
                     ("l2_trainable", dict(distance_func="l2", freeze=False))):
        m = GraphONE({k: v.clone() for k, v in banks.items()}, features_size=H, hidden_size=H, k=KG, depth=2,
                     residual=True, dropout=0, output_dropout=0, output_projection=True, **kw)
        randomize_norm_affine(m, g)
        feats = {t: torch.randn(N, H, generator=g, requires_grad=True) for t in ("ar", "lta", "pnr")}
        res, closest = m.interact(feats)
        w = {t: torch.randn(N, H, generator=g) for t in feats}
        sum((res[t] * w[t]).sum() for t in feats).backward()
        out[name] = {"kw": kw, "sd": sd_of(m), "features": {t: f.detach() for t, f in feats.items()}, "w": w,
                     "out": {t: r.detach() for t, r in res.items()},
                     "closest": {t: [c.clone() for c in cs] for t, cs in closest.items()},
                     "grad_features": {t: f.grad.clone() for t, f in feats.items()}, "grads": grads_of(m)}
    Nn, B = 12, 3
    feat = torch.randn(Nn, H, generator=g)
    batch = torch.arange(B).repeat_interleave(Nn // B)
    yb = torch.randint(0, 2, (B,), generator=g)
    t = OSCCTask(H, H, 0, 0, loss_func="bce")
    randomize_norm_affine(t, g)
    logits = t.forward_logits(t.forward_features(feat), batch)
    loss = t.compute_loss(logits, yb)
    wl = torch.randn(loss.shape, generator=g)
    (loss * wl).sum().backward()
    out["oscc_bce"] = {"sd": sd_of(t), "feat": feat, "batch": batch, "y": yb, "logits": logits.detach(), "loss": loss.detach(),
                       "w": wl, "grads": grads_of(t)}
    save("variants", out)


if __name__ == "__main__":
    golden_trn()
    golden_graph()
    golden_heads()
    golden_graphone()
    golden_build_graphone()
    golden_edges_and_loader()
    golden_mtl_train()
    golden_egopack_train()
    golden_variants()
