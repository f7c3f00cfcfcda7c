"""TRN pooling MLP over the concatenated segments of a clip node.

Mirror of reference models/temporal_pooling/trn_pooling.py:10-45 (same constructor, same
``proj.{0,1,4,5,8}`` state-dict keys).  Forward = 3 MFMA contractions (bias in the epilogue) and
2 fused LayerNorm+ReLU+dropout launches; the [N, S, F] input is read once as [N, S*F] rows."""
from __future__ import annotations

import logging

import torch.nn as nn

from ... import ops

from ..layers import Dropout, LayerNorm, Linear
from .pooling import TemporalPooling

logger = logging.getLogger(__name__)


class TRNPooling(TemporalPooling):
    def __init__(self, input_size: int = 1024, output_size: int = 1024, num_segments: int = 8,
                 hidden_size: int = 1024, dropout: float = 0.0) -> None:
        super().__init__(input_size, output_size, num_segments)
        logger.info("TRNPooling: input_size=%d hidden_size=%d output_size=%d num_segments=%d dropout=%s",
                    input_size, hidden_size, output_size, num_segments, dropout)
        self.dropout = dropout
        # container only (keeps the reference's key layout); forward below never calls proj(...)
        self.proj = nn.Sequential(
            Linear(num_segments * input_size, hidden_size), LayerNorm(hidden_size), nn.ReLU(inplace=True), Dropout(dropout),
            Linear(hidden_size, hidden_size), LayerNorm(hidden_size), nn.ReLU(inplace=True), Dropout(dropout),
            Linear(hidden_size, output_size))

    def _rows(self, x):
        if x.dim() == 3:
            if x.shape[1] != self.num_segments or x.shape[2] != self.input_size:
                raise ValueError(f"expected [N, {self.num_segments}, {self.input_size}], got {tuple(x.shape)}")
            x = x.reshape(x.shape[0], -1)  # 'bs segments h -> bs (segments h)': a view of contiguous rows
        return ops.to_act(x, lazy=True)  # no-op when the loader already delivers the mode's element type; the result goes
        # straight into the first contraction (ops.to_act: a bf16 input of the three-product mode is then never widened in memory)

    def forward(self, x, *_):
        """``x``: [N, S, F] or a list of such blocks (fused multi-task pass: rows are concatenated in
        the output of the first contraction, the inputs stay where they are)."""
        p = self.proj
        if isinstance(x, (list, tuple)):
            h0 = ops.multi_linear([self._rows(b) for b in x], p[0].weight, p[0].bias)
        else:
            h0 = p[0](self._rows(x), slab_ok=True)  # (read by the LayerNorm below and by nothing else: ops.linear)
        h = p[1](h0, relu=True, p=self.dropout)
        h = p[5](p[4](h, slab_ok=True), relu=True, p=self.dropout)
        return p[8](h)
