"""Object-state-change classification head (reference models/tasks/oscc.py:16-96): per-sequence
max pool of the node features (segment-max kernel over contiguous sequences) -> Linear(H, 2)."""
from __future__ import annotations

import logging
from typing import Dict, Literal, Optional, Tuple

import torch
import torch.nn as nn

from ... import ops
from .task import ProjectionTask, TaskLiteral, apply_classifier, build_classifier, fuse_logits

logger = logging.getLogger(__name__)


def sequence_ptr(batch) -> torch.Tensor:
    """int32 [B+1] row ranges of the sequences.  Accepts the reference's ``batch`` vector (sorted,
    contiguous sequences as PyG's collation produces) or an object carrying ``ptr``/``ptr32``."""
    if not torch.is_tensor(batch):
        ptr = getattr(batch, "ptr32", None)
        return ptr if ptr is not None else batch.ptr.to(torch.int32)
    counts = torch.bincount(batch)  # host-visible size: same sync as PyG's int(batch.max())+1
    ptr = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=batch.device)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr


class OSCCTask(ProjectionTask):
    def __init__(self, input_size: int, features_size: int, dropout: float = 0, head_dropout: float = 0,
                 loss_func: Literal["ce", "bce", "focal"] = "ce", aux_tasks: Optional[Tuple[TaskLiteral, ...]] = None,
                 average_logits: bool = False):
        super().__init__("oscc", input_size, features_size, dropout)
        logger.info("OSCC task: loss=%s dropout=%s head_dropout=%s aux=%s", loss_func, dropout, head_dropout, aux_tasks)
        self.loss_func = loss_func
        self.classifier = build_classifier(features_size, 2, head_dropout)
        if aux_tasks:
            self.aux_classifiers = nn.ModuleDict({t: build_classifier(features_size, 2, head_dropout) for t in aux_tasks})
            self.average_logits = average_logits

    def forward_logits(self, features: torch.Tensor, batch, aux_features: Optional[Dict[TaskLiteral, torch.Tensor]] = None,
                       *args, **kwargs):
        ptr = sequence_ptr(batch)
        logits = apply_classifier(self.classifier, ops.segment_max(features, ptr))
        if aux_features is not None:
            aux = [self.forward_aux_logits(f, ptr, t) for t, f in aux_features.items()]
            logits = fuse_logits(logits, aux, self.average_logits)
        return logits

    def fused_head_loss(self, features: torch.Tensor, batch, targets: torch.Tensor, smoothing: float = 0.0,
                        aux_features: Optional[Dict[TaskLiteral, torch.Tensor]] = None, aux_streams=None):
        """(loss vector, logits) of ``CrossEntropy(reduction='none', ignore_index=-1, label_smoothing)(forward_logits(features,
        batch, aux_features), targets)`` with every classifier, the logit fusion, the loss and their gradients in ONE launch behind
        the max pools (ops.linear2_ce / linear2_ce_multi), or None when it does not apply (classifier dropout active, no announced
        loss seed, many sequences, more than three auxiliary tasks): 2-logit classifiers over a few pooled rows are dozens of
        short launches of matrix work otherwise."""
        heads = [self.classifier] + ([self.aux_classifiers[t] for t in aux_features] if aux_features else [])
        feats = [features] + (list(aux_features.values()) if aux_features else [])
        if len(heads) > 4 or any(self.training and getattr(h[0], "p", 0) > 0 for h in heads):
            return None
        if not all(ops.linear2_ce_ok(int(targets.numel()), f, h[1].weight) for f, h in zip(feats, heads)):
            return None
        ptr = sequence_ptr(batch)
        # an auxiliary feature is pooled on the stream that made it (``aux_streams``: GraphONE's per-task streams), so that autograd
        # runs that pool's backward -- the head of the task's GraphONE backward chain -- there too: four pools on one stream hang
        # the three chains off consecutive nodes of that stream, and the runtime's replay then puts all three on ONE queue
        names = [None] + (list(aux_features) if aux_features else [])
        main = torch.cuda.current_stream() if features.is_cuda else None
        pooled = []
        if main is not None and not (aux_streams and any(aux_streams.get(t) is not None for t in names[1:])):
            # every input lives on this stream (the grouped GraphONE interaction): the pools of all of them as ONE launch each way
            pooled = ops.segment_max_multi(feats, ptr)
            names, feats = [], []
        for t, f in zip(names, feats):
            st = aux_streams.get(t) if (aux_streams and t is not None) else None
            if st is None or main is None:
                pooled.append(ops.segment_max(f, ptr))
                continue
            st.wait_stream(main)
            with torch.cuda.stream(st):
                p = ops.segment_max(f, ptr)
            main.wait_stream(st)
            p.record_stream(main)
            pooled.append(p)
        if len(heads) == 1:
            return ops.linear2_ce(pooled[0], heads[0][1].weight, heads[0][1].bias, targets, smoothing)
        return ops.linear2_ce_multi(pooled, [h[1].weight for h in heads], [h[1].bias for h in heads], targets, smoothing,
                                    average=bool(getattr(self, "average_logits", False)))

    def forward_aux_logits(self, features: torch.Tensor, batch, t: TaskLiteral = "ar", *args, **kwargs):
        if not hasattr(self, "aux_classifiers"):
            raise ValueError("OSCC task has no auxiliary classifiers.")
        ptr = batch if (torch.is_tensor(batch) and batch.dtype == torch.int32) else sequence_ptr(batch)
        return apply_classifier(self.aux_classifiers[t], ops.segment_max(features, ptr))

    def compute_loss(self, logits, targets):
        if self.loss_func == "ce":
            return ops.cross_entropy(logits, targets, smoothing=0.1)
        if self.loss_func == "bce":  # one_hot(targets, 2).float() targets, reduction 'none' -> [B, 2]
            return ops.onehot_bce_with_logits(logits, targets)
        if self.loss_func == "focal":  # torchvision sigmoid_focal_loss(alpha=0.5, gamma=2.0, reduction='none')
            return ops.onehot_sigmoid_focal_loss(logits, targets, alpha=0.5, gamma=2.0)
        return None  # (the reference falls through and returns None for an unknown loss_func, oscc.py:88-96)
