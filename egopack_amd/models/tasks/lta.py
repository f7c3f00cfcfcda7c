"""Long-term anticipation head (reference models/tasks/lta.py:10-74): same multi-head classifier
as AR plus ``generate_from_logits`` (K categorical samples per forecast node, validation only)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch.distributions import Categorical

from .recognition import MultiHeadTask
from .task import TaskLiteral


class LTATask(MultiHeadTask):
    def __init__(self, input_size: int, features_size: int, heads: Tuple[int, ...], dropout: float = 0,
                 head_dropout: float = 0, aux_tasks: Optional[Tuple[TaskLiteral, ...]] = None,
                 average_logits: bool = False):
        super().__init__("lta", input_size, features_size, heads, dropout, head_dropout, aux_tasks, average_logits)

    def generate_from_logits(self, logits: Tuple[torch.Tensor, ...], K=5, *args, **kwargs):
        predictions = []
        for head_logits in logits:
            dist = Categorical(logits=head_logits)
            predictions.append(torch.stack([dist.sample() for _ in range(K)], dim=1))
        return predictions, logits

    def compute_loss(self, logits, targets):
        return super().compute_loss(logits, targets)
