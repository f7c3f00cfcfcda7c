"""Validation loops (reference validate.py:13-150): eval mode, no grad, one backbone pass per batch, primary head
(optionally fused with the GraphONE aux features), per-batch mean loss, meter update.  Same signatures."""
from __future__ import annotations

from typing import List, Optional

import torch


def _eval_mode(model, primary_task, other_tasks, graphone):
    model.eval()
    for task in [primary_task, *other_tasks]:
        task.eval()
    if graphone is not None:
        graphone.eval()


# bf16 modes with a GraphONE: the backbone runs ONCE -- the forward-only precise ('bf16x3') pass tapes its nodes' results and the
# bf16 pass takes their roundings instead of launching its kernels (ops.dual_record / dual_replay, what engine.EgoPackStep's
# one-pass step does).  False: the two passes of rounds 3-5 (kept for the A/B test).
ONE_PASS = True


def _one_pass_ok(model, data) -> bool:
    from . import ops
    x = getattr(data, "x", None)
    return bool(ONE_PASS and ops.get_compute() == "bf16" and torch.is_tensor(x) and x.dtype == torch.bfloat16 and not model.training
                and not ops.graph_ln_exchange_on() and getattr(model, "stage_cut", None) is None)


def _logits(model, data, primary_task, other_tasks, graphone, late_fusion, needs_batch: bool):
    """validate.py:34-53 / :85-100 / :130-145.  ``late_fusion=False`` with a GraphONE takes the elementwise max of the
    primary and aux features before the classifier, as the reference does."""
    from . import ops
    batch = getattr(data, "batch", None)
    feat_secondary = None
    if graphone is not None and ops.get_compute() in ("bf16", "bf16_f32act"):
        # the features behind the nearest-prototype search (an index op) come from a forward-only 'bf16x3' pass: f32-grade
        # values, so that the neighbour lists are the reference's also with bf16 activations (engine.EgoPackStep)
        tape = [] if _one_pass_ok(model, data) else None
        with ops.precise_scope():
            if tape is not None:
                with ops.dual_record() as rec:
                    feat_hp = model(data)
                tape.extend(rec)
            else:
                feat_hp = model(data)
            feat_secondary = {task.name: task.forward_features(feat_hp, out_f32=True) for task in other_tasks}
        if tape:
            with ops.dual_replay(tape):  # (the bf16 features = roundings of the precise pass's results: no second backbone pass)
                feat = model(data)
        else:
            feat = model(data)
    else:
        feat = model(data)
    feat_primary = primary_task.forward_features(feat)
    if graphone is not None:
        if feat_secondary is None:
            feat_secondary = {task.name: task.forward_features(feat, out_f32=True) for task in other_tasks}  # (f32 for the search)
        feat_secondary, *_ = graphone.interact(feat_secondary)
        # post_features as the reference hands them to the meter (validate.py:43): [N, 1 + aux, H]
        feat = torch.stack([feat_primary.float(), *[f.float() for f in feat_secondary.values()]], dim=1)
        if late_fusion:
            logits = primary_task.forward_logits(features=feat_primary, batch=data if needs_batch else batch,
                                                 aux_features=feat_secondary)
        else:
            feat = feat.max(1).values
            logits = primary_task.forward_logits(feat, data) if needs_batch else primary_task.forward_logits(feat)
    else:
        feat = feat_primary
        logits = primary_task.forward_logits(feat, data) if needs_batch else primary_task.forward_logits(feat)
    return logits, feat


@torch.no_grad()
def validate(epoch, temporal_graph_model, dataloader, meter, primary_task, other_tasks: Optional[List] = None, graphone=None,
             late_fusion: bool = True, device: str = "cuda"):
    other_tasks = other_tasks or []
    _eval_mode(temporal_graph_model, primary_task, other_tasks, graphone)
    for data in dataloader:
        data = data.to(device)
        logits, feat = _logits(temporal_graph_model, data, primary_task, other_tasks, graphone, late_fusion, needs_batch=True)
        pre = data.x.mean(1) if data.x.dim() == 3 else data.x
        loss = primary_task.compute_loss(logits, data.y).mean()
        meter.update(logits, data.y, loss, pre, feat)


@torch.no_grad()
def validate_lta(temporal_graph_model, dataloader, meter, primary_task, other_tasks: Optional[List] = None, graphone=None,
                 late_fusion: bool = False, device: str = "cuda"):
    other_tasks = other_tasks or []
    _eval_mode(temporal_graph_model, primary_task, other_tasks, graphone)
    for data in dataloader:
        data = data.to(device)
        logits, _ = _logits(temporal_graph_model, data, primary_task, other_tasks, graphone, late_fusion, needs_batch=True)
        predictions, logits = primary_task.generate_from_logits(logits)
        loss = primary_task.compute_loss(logits, data.y).mean()
        meter.update(logits, data.y, predictions, loss)


@torch.no_grad()
def validate_pnr(temporal_graph_model, dataloader, meter, primary_task, other_tasks: Optional[List] = None, graphone=None,
                 late_fusion: bool = False, device: str = "cuda"):
    other_tasks = other_tasks or []
    _eval_mode(temporal_graph_model, primary_task, other_tasks, graphone)
    for data in dataloader:
        data = data.to(device)
        logits, _ = _logits(temporal_graph_model, data, primary_task, other_tasks, graphone, late_fusion, needs_batch=False)
        loss = primary_task.compute_loss(logits, data.y)  # (the reference passes y.float(); the BCE kernel converts)
        # (the batch object carries the int32 sequence pointer the collation built; a plain ``batch`` vector works too)
        meter.update(logits, data.y, data if getattr(data, "ptr32", None) is not None else data.batch,
                     data.start_frame, data.end_frame, data.pnr_frame, loss)
