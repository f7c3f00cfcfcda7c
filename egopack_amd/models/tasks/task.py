"""Shared projection head of every task (reference models/tasks/task.py:8-26).

``net`` keeps the reference's Sequential layout (keys net.{1,2,4}; ``net[-1].out_features`` is read
by the prototype builder) but the forward is 2 MFMA contractions + 1 fused LayerNorm+ReLU launch."""
from __future__ import annotations

from typing import List, Literal

import torch
import torch.nn as nn

from ..layers import Dropout, LayerNorm, Linear

TaskLiteral = Literal["ar", "oscc", "lta", "pnr", "ant"]


class ProjectionTask(torch.nn.Module):
    def __init__(self, name: str, input_size: int, features_size: int = 1024, dropout: float = 0):
        super().__init__()
        self.name, self.input_size, self.features_size = name, input_size, features_size
        self.net = nn.Sequential(Dropout(dropout), Linear(input_size, features_size), LayerNorm(features_size),
                                 nn.ReLU(), Linear(features_size, features_size))

    def forward_features(self, x: torch.Tensor, *args, out_f32: bool = False, **kwargs) -> torch.Tensor:
        """``out_f32`` (not a reference argument): keep the last contraction's f32 accumulators as the output whatever the
        activation storage type -- the GraphONE prototype search then ranks f32 values, not their bf16 roundings."""
        from ... import ops
        n = self.net
        return n[4](n[2](n[1](n[0](ops.to_act(x))), relu=True), out_f32=out_f32)

    def configure_optimizers(self, _):
        return self.parameters()


def build_classifier(features_size: int, out: int, head_dropout: float) -> nn.Sequential:
    return nn.Sequential(Dropout(head_dropout), Linear(features_size, out))


def apply_classifier(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Dropout -> Linear; logits are kept in f32 whatever the activation element type."""
    return seq[1](seq[0](x), out_f32=True)


def fuse_logits(primary: torch.Tensor, aux: List[torch.Tensor], average: bool) -> torch.Tensor:
    """stack([primary, *aux]).sum(0) / .mean(0) of the reference (recognition.py:53-57) without the
    stacked copy: a running axpby over the (small) logit tensors."""
    from ... import ops
    return ops.sum_tensors([primary, *aux], scale=(1.0 / (1 + len(aux))) if average else 1.0)
