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

    def fused_head_loss(self, features: torch.Tensor, batch, targets: torch.Tensor, smoothing: float = 0.0):
        """(loss vector, logits) of ``CrossEntropy(reduction='none', ignore_index=-1)(forward_logits(features, batch), targets)``
        with the classifier, the loss and their gradients in ONE launch behind the max pool (ops.linear2_ce), or None when it
        does not apply (classifier dropout active, no announced loss seed, many sequences): a 2-logit classifier over a few
        pooled rows is eleven short launches of matrix work otherwise."""
        drop, lin = self.classifier[0], self.classifier[1]
        if (self.training and getattr(drop, "p", 0) > 0) or not ops.linear2_ce_ok(int(targets.numel()), features, lin.weight):
            return None
        pooled = ops.segment_max(features, sequence_ptr(batch))
        return ops.linear2_ce(pooled, lin.weight, lin.bias, targets, smoothing)

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
