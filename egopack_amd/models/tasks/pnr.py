"""Point-of-no-return head (reference models/tasks/pnr.py:12-83): one logit per clip node,
BCE-with-logits against the one-hot PNR position."""
from __future__ import annotations

import logging
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from ... import ops
from .task import ProjectionTask, TaskLiteral, apply_classifier, build_classifier, fuse_logits

logger = logging.getLogger(__name__)


class PNRTask(ProjectionTask):
    def __init__(self, input_size: int, features_size: int, dropout: float = 0, head_dropout: float = 0,
                 aux_tasks: Optional[Tuple[TaskLiteral, ...]] = None, average_logits: bool = False):
        super().__init__("pnr", input_size, features_size, dropout)
        self.classifier = build_classifier(features_size, 1, head_dropout)
        if aux_tasks:
            self.aux_classifiers = nn.ModuleDict({t: build_classifier(features_size, 1, head_dropout) for t in aux_tasks})
            self.average_logits = average_logits

    def forward(self, x: torch.Tensor, *args, **kwargs):
        features = self.forward_features(x)
        return apply_classifier(self.classifier, features).squeeze(), features

    def forward_logits(self, features: torch.Tensor, aux_features: Optional[Dict[TaskLiteral, torch.Tensor]] = None,
                       *args, **kwargs):
        logits = apply_classifier(self.classifier, features)  # [N, 1]
        if aux_features is not None:
            aux = [self.forward_aux_logits(f, t) for t, f in aux_features.items()]
            logits = fuse_logits(logits, aux, self.average_logits)
        return logits.squeeze()

    def forward_aux_logits(self, features: torch.Tensor, t: TaskLiteral = "ar", *args, **kwargs):
        if not hasattr(self, "aux_classifiers"):
            raise ValueError("PNR task has no auxiliary classifiers.")
        return apply_classifier(self.aux_classifiers[t], features)

    def fused_head_loss(self, features: torch.Tensor, targets: torch.Tensor):
        """(loss vector, logits) of ``BCEWithLogits(forward_logits(features), targets)`` in ONE row pass that also emits the
        gradients (ops.linear1_bce), or None when it does not apply (classifier dropout active, no announced loss seed):
        a one-logit classifier is a row reduction, not matrix work."""
        drop, lin = self.classifier[0], self.classifier[1]
        if (self.training and getattr(drop, "p", 0) > 0) or not ops.linear1_bce_ok(features, lin.weight):
            return None
        return ops.linear1_bce(features, lin.weight, lin.bias, targets)

    def compute_loss(self, logits: torch.Tensor, targets: torch.Tensor):
        return ops.bce_with_logits(logits, targets)
