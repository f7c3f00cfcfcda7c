"""Loss selection wrapper (reference criterion/wrapper.py:11-82)."""
from __future__ import annotations

import logging
from typing import Tuple

import torch

from . import ops

logger = logging.getLogger(__name__)


class CrossEntropyNone(torch.nn.Module):
    """nn.CrossEntropyLoss(reduction='none', ignore_index=-1[, label_smoothing]) on the HIP path."""

    def __init__(self, ignore_index: int = -1, label_smoothing: float = 0.0):
        super().__init__()
        if ignore_index != -1:
            raise ValueError("the hot path uses ignore_index=-1")
        self.label_smoothing = label_smoothing

    def forward(self, logits, target):
        return ops.cross_entropy(logits, target, self.label_smoothing)


class BCEWithLogitsNone(torch.nn.Module):
    """nn.BCEWithLogitsLoss(reduction='none'); the caller passes ``y.float()`` as the reference does."""

    def forward(self, logits, target):
        return ops.bce_with_logits(logits, target)


class MetricSelectorWrapper(torch.nn.Module):
    """Apply a per-head criterion to the heads selected by the dataset's label structure and sum the
    per-head loss vectors.  Needs only ``dataset.has_joint_label`` and ``dataset.num_labels``."""

    def __init__(self, criterion: torch.nn.Module, dataset, joint_label_training: bool = False) -> None:
        super().__init__()
        if not dataset.has_joint_label and joint_label_training:
            logger.warning("The flag join_labels is set to True but the dataset has no joint label")
            joint_label_training = False
        self.criterion, self.dataset, self.joint_label = criterion, dataset, joint_label_training

    def select(self, logits: Tuple[torch.Tensor, ...], ground_truths: torch.Tensor):
        """(logits of the heads that count, their label columns, label smoothing): what ``forward`` hands to the cross
        entropy (the engine batches the cross entropies of several tasks into one launch from these)."""
        if len(logits) != ground_truths.shape[1]:
            raise ValueError("The number of predictions must match the number of ground truth labels")
        if self.dataset.has_joint_label:
            heads = [len(logits) - 1] if self.joint_label else list(range(self.dataset.num_labels - 1))
        else:
            heads = list(range(self.dataset.num_labels))
        smoothing = getattr(self.criterion, "label_smoothing", 0.0)
        if heads == list(range(ground_truths.shape[1])):
            return tuple(logits), ground_truths, smoothing  # all heads: the common case on the path
        return tuple(logits[h] for h in heads), ground_truths[:, heads].contiguous(), smoothing

    def forward(self, logits: Tuple[torch.Tensor, ...], ground_truths: torch.Tensor) -> torch.Tensor:
        sel, gt, smoothing = self.select(logits, ground_truths)
        return ops.cross_entropy(sel, gt, smoothing)  # one fused per-row sum over the heads
