"""Multi-head recognition task (AR): verb and noun classifiers over the projected clip feature.

Mirror of reference models/tasks/recognition.py:10-72 (constructor, ``classifiers`` /
``aux_classifiers`` layout, ``forward_logits`` / ``forward_aux_logits`` / ``compute_loss``)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from ... import ops
from .task import ProjectionTask, TaskLiteral, apply_classifier, build_classifier, fuse_logits


class MultiHeadTask(ProjectionTask):
    """Common body of the AR and LTA heads (they differ in name and in LTA's sampler)."""

    def __init__(self, name: str, input_size: int, features_size: int, heads: Tuple[int, ...], dropout: float = 0,
                 head_dropout: float = 0, aux_tasks: Optional[Tuple[TaskLiteral, ...]] = None,
                 average_logits: bool = False):
        super().__init__(name, input_size, features_size, dropout)
        self.heads = tuple(heads)
        self.classifiers = self._build_classifier(head_dropout, heads)
        key = object()  # (identity of this bank; a deep copy of the task gets its own)
        for h, c in enumerate(self.classifiers):  # one bank: optim.FlatAdam lays their padded slots out as one matrix
            c[1].weight._egk_bank, c[1].bias._egk_bank = (key, "w", h), (key, "b", h)
        if aux_tasks:
            self.aux_classifiers = nn.ModuleDict({t: self._build_classifier(head_dropout, heads) for t in aux_tasks})
            self.average_logits = average_logits

    def _build_classifier(self, head_dropout, heads) -> nn.ModuleList:
        return nn.ModuleList([build_classifier(self.features_size, h, head_dropout) for h in heads])

    def forward_logits(self, features: torch.Tensor, batch: Optional[torch.Tensor] = None,
                       aux_features: Optional[Dict[TaskLiteral, torch.Tensor]] = None, *args, **kwargs):
        bank = getattr(self.classifiers[0][1].weight, "_egk_bank_views", None)
        if bank is not None and features.is_cuda and not (self.training and any(c[0].p > 0 for c in self.classifiers)):
            logits = ops.classifier_bank(features, self.classifiers[0][1].weight, bank)  # one contraction for all heads
        else:
            logits = tuple(apply_classifier(c, features) for c in self.classifiers)
        if aux_features is not None:
            aux = [self.forward_aux_logits(f, t) for t, f in aux_features.items()]
            logits = tuple(fuse_logits(p, [a[h] for a in aux], self.average_logits) for h, p in enumerate(logits))
        return logits

    def forward_aux_logits(self, features: torch.Tensor, t: TaskLiteral = "ar", *args, **kwargs):
        return tuple(apply_classifier(c, features) for c in self.aux_classifiers[t])

    def compute_loss(self, logits: Tuple[torch.Tensor, ...], targets: torch.Tensor, return_separate_losses: bool = False):
        """sum over heads of CrossEntropy(reduction='none', ignore_index=-1)."""
        total = ops.cross_entropy(tuple(logits), targets)
        if return_separate_losses:
            return total, tuple(ops.cross_entropy(l, targets[:, i].contiguous()) for i, l in enumerate(logits))
        return total


class RecognitionTask(MultiHeadTask):
    def __init__(self, input_size: int, features_size: int, heads: Tuple[int, ...], dropout: float = 0,
                 head_dropout: float = 0, aux_tasks: Optional[Tuple[TaskLiteral, ...]] = None,
                 average_logits: bool = False):
        super().__init__("ar", input_size, features_size, heads, dropout, head_dropout, aux_tasks, average_logits)
