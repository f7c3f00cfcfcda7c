"""Base class of the temporal pooling modules (reference models/temporal_pooling/pooling.py:9-45).

Only the constructor surface used on the executed path is kept: TRNPooling passes no ``encoding``
(trn_pooling.py:20), so the positional / temporal / learnt encodings of the base class never run."""
from __future__ import annotations

import torch


class TemporalPooling(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, num_segments: int, encoding=None, encoding_level: str = "frame"):
        super().__init__()
        if encoding is not None:
            raise NotImplementedError("temporal-pooling encodings are not on the hot path (TRNPooling uses none)")
        self.input_size, self.output_size, self.num_segments = input_size, output_size, num_segments
        self.encoding_level, self.encoding, self.encoding_mlp = encoding_level, None, None

    def apply_positional_embedding(self, x, batch, pos):
        return x

    def forward(self, x, batch, pos):
        raise NotImplementedError("TemporalPooling.forward is not implemented")
