"""Oracle restatement of the reference's own hot-path modules -- TEST INFRASTRUCTURE ONLY.

Functional (state-dict driven) plain-PyTorch fp32 restatement of SURVEY.md section 8(a) rows
a1..a19.  Every function takes the tensors of a reference-layout ``state_dict`` (same key
names as the reference checkpoints) so that one set of parameters can be pushed
through (i) the imported reference modules (oracle/make_golden.py), (ii) this oracle and
(iii) the HIP product path, and compared.

PINNED: oracle/make_golden.py runs the reference's own Python on the same inputs and
tests/test_oracle_golden.py checks this file against those stored outputs.
"""
from __future__ import annotations

from math import floor
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import pyg_ops as P
from . import storage as S  # (identity hooks unless the bf16 storage model is switched on: oracle/storage.py)

SD = Mapping[str, torch.Tensor]


def _sub(sd: SD, prefix: str) -> Dict[str, torch.Tensor]:
    """View of the entries of ``sd`` below ``prefix`` with the prefix stripped."""
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def _dropout(x: torch.Tensor, p: float, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Dropout in training with an explicit keep-mask (1 = keep); identity if mask is None."""
    if mask is None or p == 0.0:
        return x
    return x * mask.to(x.dtype) / (1.0 - p)


# --------------------------------------------------------------------------------------
# a2  TRNPooling  (reference models/temporal_pooling/trn_pooling.py:28-45)
# --------------------------------------------------------------------------------------
def trn_pooling(sd: SD, x: torch.Tensor, dropout: float = 0.0,
                masks: Optional[Sequence[Optional[torch.Tensor]]] = None) -> torch.Tensor:
    """rearrange 'bs segments h -> bs (segments h)' then
    Linear -> LayerNorm -> ReLU -> Dropout -> Linear -> LayerNorm -> ReLU -> Dropout -> Linear.
    ``sd`` keys: proj.{0,1,4,5,8}.{weight,bias}.  ``masks`` = keep-masks of the two dropouts
    (training mode) or None (eval)."""
    m0, m1 = (masks if masks is not None else (None, None))
    h = x.reshape(x.shape[0], -1)
    h = S.act(F.linear(h, S.weight(sd["proj.0.weight"]), sd["proj.0.bias"]))
    h = F.layer_norm(h, h.shape[-1:], sd["proj.1.weight"], sd["proj.1.bias"], 1e-5)
    h = S.act(_dropout(F.relu(h), dropout, m0))
    h = S.act(F.linear(h, S.weight(sd["proj.4.weight"]), sd["proj.4.bias"]))
    h = F.layer_norm(h, h.shape[-1:], sd["proj.5.weight"], sd["proj.5.bias"], 1e-5)
    h = S.act(_dropout(F.relu(h), dropout, m1))
    return S.act(F.linear(h, S.weight(sd["proj.8.weight"]), sd["proj.8.bias"]))


# --------------------------------------------------------------------------------------
# a1/a3..a6  Graph.forward  (reference models/graph.py:53-65)
# --------------------------------------------------------------------------------------
def graph_forward(sd: SD, x: torch.Tensor, pos: torch.Tensor, edge_index: torch.Tensor,
                  depth: int = 3, trn_dropout: float = 0.0,
                  trn_masks: Optional[Sequence[Optional[torch.Tensor]]] = None,
                  pre_dropout: float = 0.0, pre_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x = pre_dropout(x); x = TRNPooling(x); x = x + net(x + PE(pos), edge_index) where
    net = depth x [SAGEConv(H,H,project=True,mean) -> graph LayerNorm -> LeakyReLU(0.2)]
    -> Linear(H,H).  State-dict keys follow PyG's ``module_<i>`` naming."""
    x = _dropout(x, pre_dropout, pre_mask)
    x = trn_pooling(_sub(sd, "temporal_pooling."), x, trn_dropout, trn_masks)
    freq = sd["positional_encoding.frequency"]
    h = S.act(x + P.positional_encoding(pos, freq))
    for d in range(depth):
        c, n = f"net.module_{3 * d}.", f"net.module_{3 * d + 1}."
        h = S.act(P.sage_conv(h, edge_index,
                              sd[c + "lin_l.weight"], sd[c + "lin_l.bias"], sd[c + "lin_r.weight"],
                              sd[c + "lin.weight"], sd[c + "lin.bias"], aggr="mean"))
        h = P.graph_layer_norm(h, sd[n + "weight"], sd[n + "bias"])
        h = S.act(F.leaky_relu(h, 0.2))
    last = f"net.module_{3 * depth}."
    h = F.linear(h, S.weight(sd[last + "weight"]), sd[last + "bias"])
    return S.act(x + h)


# --------------------------------------------------------------------------------------
# a7..a10  task heads  (reference models/tasks/*.py)
# --------------------------------------------------------------------------------------
def projection_features(sd: SD, x: torch.Tensor, dropout: float = 0.0,
                        mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ProjectionTask.net: Dropout -> Linear -> LayerNorm -> ReLU -> Linear
    (reference models/tasks/task.py:17-26).  Keys net.{1,2,4}.{weight,bias}."""
    h = _dropout(x, dropout, mask)
    if mask is not None and dropout > 0:
        h = S.act(h)
    h = S.act(F.linear(h, S.weight(sd["net.1.weight"]), sd["net.1.bias"]))
    h = S.act(F.relu(F.layer_norm(h, h.shape[-1:], sd["net.2.weight"], sd["net.2.bias"], 1e-5)))
    return S.act(F.linear(h, S.weight(sd["net.4.weight"]), sd["net.4.bias"]))


def _fuse(primary: torch.Tensor, aux: List[torch.Tensor], average: bool) -> torch.Tensor:
    stack = torch.stack([primary, *aux])
    return stack.mean(0) if average else stack.sum(0)


def multihead_logits(sd: SD, features: torch.Tensor, n_heads: int,
                     aux_features: Optional[Mapping[str, torch.Tensor]] = None,
                     average_logits: bool = False) -> Tuple[torch.Tensor, ...]:
    """RecognitionTask / LTATask.forward_logits (reference recognition.py:39-59, lta.py:39-58):
    per head Linear(H,C); with aux features each aux task's own classifier bank is applied to
    that task's GraphONE feature and the per-head logits are summed (or averaged).
    (head dropout is identity in eval / p=0, which is what the oracle is used with.)"""
    logits = [S.grad(F.linear(features, S.weight(sd[f"classifiers.{h}.1.weight"]), sd[f"classifiers.{h}.1.bias"]))
              for h in range(n_heads)]
    if aux_features is not None:
        fused = []
        for h in range(n_heads):
            aux = [S.grad(F.linear(f, S.weight(sd[f"aux_classifiers.{t}.{h}.1.weight"]), sd[f"aux_classifiers.{t}.{h}.1.bias"]))
                   for t, f in aux_features.items()]
            fused.append(_fuse(logits[h], aux, average_logits))
        logits = fused
    return tuple(logits)


def oscc_logits(sd: SD, features: torch.Tensor, batch: torch.Tensor,
                aux_features: Optional[Mapping[str, torch.Tensor]] = None,
                average_logits: bool = False, num_graphs: Optional[int] = None) -> torch.Tensor:
    """OSCCTask.forward_logits (reference oscc.py:65-86): global_max_pool then Linear(H,2)."""
    pooled = S.act(P.global_max_pool(features, batch, num_graphs))
    logits = S.grad(F.linear(pooled, S.weight(sd["classifier.1.weight"]), sd["classifier.1.bias"]))
    if aux_features is not None:
        aux = [S.grad(F.linear(S.act(P.global_max_pool(f, batch, num_graphs)),
                               S.weight(sd[f"aux_classifiers.{t}.1.weight"]), sd[f"aux_classifiers.{t}.1.bias"]))
               for t, f in aux_features.items()]
        logits = _fuse(logits, aux, average_logits)
    return logits


def pnr_logits(sd: SD, features: torch.Tensor,
               aux_features: Optional[Mapping[str, torch.Tensor]] = None,
               average_logits: bool = False) -> torch.Tensor:
    """PNRTask.forward_logits (reference pnr.py:62-80): Linear(H,1).squeeze(); aux logits are
    stacked with logits.unsqueeze(1) and summed / averaged, then squeezed again."""
    logits = S.grad(F.linear(features, S.weight(sd["classifier.1.weight"]), sd["classifier.1.bias"])).squeeze()
    if aux_features is not None:
        aux = [S.grad(F.linear(f, S.weight(sd[f"aux_classifiers.{t}.1.weight"]), sd[f"aux_classifiers.{t}.1.bias"]))
               for t, f in aux_features.items()]
        logits = _fuse(logits.unsqueeze(1), aux, average_logits)
    return logits.squeeze()


# --------------------------------------------------------------------------------------
# a11  losses
# --------------------------------------------------------------------------------------
def multihead_ce(logits: Sequence[torch.Tensor], y: torch.Tensor) -> torch.Tensor:
    """MetricSelectorWrapper.forward for a dataset without joint label (reference
    criterion/wrapper.py:44-82) == RecognitionTask/LTATask.compute_loss (recognition.py:61-69):
    sum over heads of CrossEntropyLoss(reduction='none', ignore_index=-1); ignored rows give 0
    and still count in the caller's ``.mean()`` (main_temporal.py:99)."""
    if len(logits) != y.shape[1]:
        raise ValueError("The number of predictions must match the number of ground truth labels")
    return torch.stack([F.cross_entropy(l, y[:, i], ignore_index=-1, reduction="none")
                        for i, l in enumerate(logits)]).sum(0)


def oscc_loss(logits: torch.Tensor, y: torch.Tensor, kind: str = "ce", smoothing: float = 0.1) -> torch.Tensor:
    """OSCCTask.compute_loss (reference oscc.py:88-96).  MTL pre-training uses a plain
    CrossEntropyLoss instead (main_temporal.py:291): call with smoothing=0."""
    if kind == "ce":
        return F.cross_entropy(logits, y, ignore_index=-1, reduction="none", label_smoothing=smoothing)
    if kind == "bce":
        return F.binary_cross_entropy_with_logits(logits, F.one_hot(y, 2).float(), reduction="none")
    if kind == "focal":  # torchvision.ops.sigmoid_focal_loss(alpha=0.5, gamma=2, reduction='none')
        t = F.one_hot(y, 2).float()
        p = torch.sigmoid(logits)
        ce = F.binary_cross_entropy_with_logits(logits, t, reduction="none")
        p_t = p * t + (1 - p) * (1 - t)
        return (0.5 * t + 0.5 * (1 - t)) * ce * (1 - p_t) ** 2.0
    raise ValueError(kind)


def pnr_loss(logits: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """BCEWithLogitsLoss(reduction='none') on y.float() (reference main_temporal.py:123,298; pnr.py:82-83)."""
    return F.binary_cross_entropy_with_logits(logits, y.float(), reduction="none")


# --------------------------------------------------------------------------------------
# a12..a14  GraphONE  (reference models/graphONE/graphONE.py)
# --------------------------------------------------------------------------------------
def cos_dissimilarity(g1: torch.Tensor, g2: torch.Tensor) -> torch.Tensor:
    """reference graphONE.py:148-151."""
    g1 = g1 / g1.norm(dim=1, keepdim=True)
    g2 = g2 / g2.norm(dim=1, keepdim=True)
    return 1 - torch.mm(g1, g2.T)


@torch.no_grad()
def compute_edges(features: torch.Tensor, bank: torch.Tensor, k: int,
                  distance_func: str = "cosine") -> Tuple[torch.Tensor, torch.Tensor]:
    """GraphONE.__compute_edges (reference graphONE.py:119-141): k nearest prototypes of every
    node by full argsort of the distance rows; edges = [closest.flatten(); K + node index
    repeated k times].  Returns (edges [2, N*k], closest [N, k]); the softmax weights / entropy
    the reference also computes are discarded by its caller (:102) and omitted here."""
    K, B = bank.shape[0], features.shape[0]
    if distance_func == "cosine":
        dist = cos_dissimilarity(features, bank)
    elif distance_func == "l2":
        dist = torch.cdist(features, bank, p=2, compute_mode="donot_use_mm_for_euclid_dist") / 4096
    else:
        raise ValueError(f"Unknown distance function: {distance_func}")
    closest = dist.argsort(dim=-1, descending=False)[:, :k]
    tgt = torch.arange(K, K + B, device=bank.device).repeat_interleave(closest.shape[1])
    return torch.stack([closest.flatten(), tgt]), closest


def graphone_task_interaction(sd: SD, task: str, features: torch.Tensor, k: int, depth: int,
                              residual: bool = False, distance_func: str = "cosine",
                              closest_override: Optional[torch.Tensor] = None,
                              ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """GraphONE.__task_interaction (reference graphONE.py:87-117).  Edges are always computed
    from the ORIGINAL features and bank (features_match is never reassigned), the SAGE stage runs
    over cat([bank, features]) with remaining self loops and only the last N rows are kept.
    ``closest_override`` [N, k] (tests only): use these prototype indices instead of the searched ones,
    to measure what a different neighbour selection does to the outputs."""
    bank = sd[f"embeddings.{task}.weight"]
    f0 = features
    n = features.shape[0]
    assignments: List[torch.Tensor] = []
    for d in range(depth):
        edges, closest = compute_edges(f0, bank, k, distance_func)
        if closest_override is not None:
            closest = closest_override
            tgt = torch.arange(bank.shape[0], bank.shape[0] + n).repeat_interleave(k)
            edges = torch.stack([closest.flatten(), tgt])
        assignments.append(closest[:, 0])
        graph = torch.cat([bank, features], dim=0)
        edges = P.add_remaining_self_loops(edges, graph.shape[0])
        s = f"conv_stages.{task}.{d}."
        h = S.act(P.sage_conv(graph, edges, sd[s + "module_0.lin_l.weight"], None, sd[s + "module_0.lin_r.weight"], aggr="max"))
        h = S.act(F.relu(F.layer_norm(h, h.shape[-1:], sd[s + "module_1.weight"], sd[s + "module_1.bias"], 1e-5)))
        h = F.linear(h, S.weight(sd[s + "module_3.weight"]), sd[s + "module_3.bias"])
        features = S.act(h[-n:] + features if residual else h[-n:])
    return features, assignments


def graphone_interact(sd: SD, features: Mapping[str, torch.Tensor], k: int, depth: int,
                      residual: bool = False, distance_func: str = "cosine",
                      closest_override: Optional[Mapping[str, torch.Tensor]] = None):
    """GraphONE.interact (reference graphONE.py:76-85)."""
    out, closest = {}, {}
    for task, f in features.items():
        out[task], closest[task] = graphone_task_interaction(
            sd, task, f, k, depth, residual, distance_func,
            None if closest_override is None else closest_override[task])
    return out, closest


# --------------------------------------------------------------------------------------
# a15  prototype bank builder (reference graphone.py:17-63)
# --------------------------------------------------------------------------------------
@torch.no_grad()
def build_graphone(backbone_sd: SD, task_sds: Mapping[str, SD], batches: Sequence[P.OData],
                   n_classes: Tuple[int, int], depth: int = 3) -> Dict[str, torch.Tensor]:
    """Eval-mode pass: per batch keep nodes with y[:,0] != -1, per task scatter-sum the task
    features by label = verb*|nouns|+noun into a float64 bank, bincount, mean over seen rows,
    drop unseen rows, cast to fp32."""
    size = n_classes[0] * n_classes[1]
    feat_size = next(iter(task_sds.values()))["net.4.weight"].shape[0]
    banks = {t: torch.zeros((size, feat_size), dtype=torch.float64) for t in task_sds}
    all_labels = []
    for data in batches:
        feat = graph_forward(backbone_sd, data.x, data.pos, data.edge_index, depth)
        keep = data.y[:, 0] != -1
        feat, y = feat[keep], data.y[keep]
        for t, tsd in task_sds.items():
            tf = projection_features(tsd, feat)
            labels = y[:, 0] * n_classes[1] + y[:, 1]
            all_labels.append(labels)
            banks[t] = banks[t] + P.scatter_sum(tf, labels, size)  # fp64 + fp32 -> fp64
    count = torch.cat(all_labels).bincount(minlength=size).float()
    seen = count > 0
    return {t: (b[seen] / count[seen, None]).float() for t, b in banks.items()}


# --------------------------------------------------------------------------------------
# a17  edge builders (integer, bit-exact as a set)
# --------------------------------------------------------------------------------------
def temporal_radius_edges(pos: torch.Tensor, k: int) -> torch.Tensor:
    """RadiusGraph(r=k+0.5, loop=False) on one sample (reference main_temporal.py:168)."""
    return P.radius_graph(pos, k + 0.5, None, loop=False, max_num_neighbors=32)


def lta_temporal_connectivity(pos: torch.Tensor, y: torch.Tensor, r: float) -> torch.Tensor:
    """LTATemporalConnectivity.__call__ (reference lta_temp_connectivity.py:30-56): radius band
    plus edges from the last floor(r) input clips to every forecast clip, then coalesce.
    Keeps the reference's quirk: forecast clips are counted with ``y[:,0] > 0`` (a verb label 0
    is NOT counted)."""
    band = P.radius_graph(pos, r, None, loop=False, max_num_neighbors=32)
    n_in = int((y[:, 0] == -1).sum())
    n_f = int((y[:, 0] > 0).sum())
    import math
    lo = max(math.ceil(n_in - r), 0)
    src = torch.arange(lo, n_in, dtype=torch.long).repeat_interleave(n_f)
    tgt = torch.arange(n_in, n_in + n_f, dtype=torch.long).repeat(min(floor(r), n_in))
    ei = torch.cat([torch.stack([src, tgt]), band], dim=-1)
    return P.coalesce(ei, pos.shape[0])


# --------------------------------------------------------------------------------------
# a16  step objectives
# --------------------------------------------------------------------------------------
def mtl_objective(backbone_sd: SD, task_sds: Mapping[str, SD], batches: Mapping[str, P.OData],
                  weights: Mapping[str, float], depth: int = 3, n_heads: int = 2, trn_dropout: float = 0.0,
                  trn_masks: Optional[Mapping[str, Sequence[torch.Tensor]]] = None):
    """main_temporal.train body (reference main_temporal.py:87-128): one backbone forward per
    enabled task batch, its head, its loss vector, ``weight * loss.mean()`` summed.
    Task order ar, lta, oscc, pnr as in the reference.  Returns (total, per-task dict of
    (logits, loss_vector)).  ``trn_masks`` {task: (mask0, mask1)}: keep masks of the temporal pooling's two dropouts
    (training mode with ``trn_dropout``), None = eval."""
    total, detail = [], {}
    for t in ("ar", "lta", "oscc", "pnr"):
        if t not in batches or weights.get(t, 0) <= 0:
            continue
        d = batches[t]
        feat = graph_forward(backbone_sd, d.x, d.pos, d.edge_index, depth, trn_dropout=trn_dropout,
                             trn_masks=None if trn_masks is None else trn_masks[t])
        f = projection_features(task_sds[t], feat)
        if t in ("ar", "lta"):
            logits = multihead_logits(task_sds[t], f, n_heads)
            loss = multihead_ce(logits, d.y)
        elif t == "oscc":
            logits = oscc_logits(task_sds[t], f, d.batch, num_graphs=getattr(d, "num_graphs", None))
            loss = oscc_loss(logits, d.y, "ce", smoothing=0.0)
        else:
            logits = pnr_logits(task_sds[t], f)
            loss = pnr_loss(logits, d.y)
        detail[t] = (logits, loss)
        total.append(weights[t] * loss.mean())
    return torch.stack(total).sum(), detail


def egopack_task_loss(primary: str, task_sds: Mapping[str, SD], graphone_sd: SD, feat: torch.Tensor,
                      batch: torch.Tensor, y: torch.Tensor, others: Sequence[str],
                      k: int, depth: int, residual: bool, average_logits: bool,
                      oscc_kind: str = "ce", n_heads: int = 2, num_graphs: Optional[int] = None,
                      closest_override: Optional[Mapping[str, torch.Tensor]] = None,
                      aux_source: Optional[torch.Tensor] = None):
    """main_egopack.train_step_task (reference main_egopack.py:45-61) with late_fusion=True:
    primary features; aux features = GraphONE.interact on the DETACHED projections of the other
    tasks; fused logits; primary.compute_loss.
    ``aux_source`` (bf16 storage model only): the backbone features the detached auxiliary projections are taken from -- the
    product computes them in a separate f32-grade pass (they feed an index op), so under the storage model they come from
    f32 features through f32 projections and are rounded once, when they enter the GraphONE stages."""
    f_primary = projection_features(task_sds[primary], feat)
    if aux_source is not None:
        with S.bf16_storage(False), torch.no_grad():
            aux_in = {t: projection_features(task_sds[t], aux_source) for t in others}
            if closest_override is None:  # the search ranks the f32-grade values, not their bf16 roundings
                closest_override = {t: compute_edges(aux_in[t], graphone_sd[f"embeddings.{t}.weight"], k)[1] for t in others}
        aux_in = {t: S.act(a) for t, a in aux_in.items()}
    else:
        aux_in = {t: projection_features(task_sds[t], feat).detach() for t in others}
    aux, closest = graphone_interact(graphone_sd, aux_in, k, depth, residual, closest_override=closest_override)
    sd = task_sds[primary]
    if primary in ("ar", "lta"):
        logits = multihead_logits(sd, f_primary, n_heads, aux, average_logits)
        loss = multihead_ce(logits, y)
    elif primary == "oscc":
        logits = oscc_logits(sd, f_primary, batch, aux, average_logits, num_graphs)
        loss = oscc_loss(logits, y, oscc_kind)
    else:
        logits = pnr_logits(sd, f_primary, aux, average_logits)
        loss = pnr_loss(logits, y)
    return loss, logits, aux, closest
