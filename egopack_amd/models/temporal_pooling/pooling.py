"""Base class of the temporal pooling modules (reference models/temporal_pooling/pooling.py:9-83).

Same constructor, attributes and state-dict keys (``encoding`` / ``encoding.frequency`` / ``encoding.weight``,
``encoding_mlp.{weight,bias}``) and the reference's ``apply_positional_embedding``: an optional encoding of the segment index
('frame' level: every action's segments get the same S rows) or of the action's position ('action' level: all segments of
an action share the row of its ``pos``), passed through ``encoding_mlp`` and added to the input.  TRNPooling -- the only
subclass on the executed path -- passes no encoding (trn_pooling.py:20), so none of this runs in the benchmarked step; it
is here so that a configuration that does ask for an encoding gets the reference's arithmetic instead of an error:
the MLP is an ``egk_gemm`` launch, the sinusoidal rows come from ``egk_pe_add``; the final broadcast add of an [S, F] (or
[N, 1, F]) encoding to the [N, S, F] input is left to torch (one elementwise launch).

PyG's two encodings are restated from their published definitions (the package is absent):
``PositionalEncoding(C)``: cat[sin(p f), cos(p f)], f = logspace(0, 1, C/2, base 1e-4);
``TemporalEncoding(C)``: sqrt(1 / C) * cos(p w), w = 1 / 10^linspace(0, 9, C)."""
from __future__ import annotations

import logging
import math
from typing import Optional

import torch

from ... import ops
from ..layers import Linear, PositionalEncoding

logger = logging.getLogger(__name__)


class TemporalEncoding(torch.nn.Module):
    """gnn.TemporalEncoding(C): buffer ``weight`` = 1 / 10^linspace(0, 9, C); forward sqrt(1 / C) * cos(x * weight)."""

    def __init__(self, out_channels: int):
        super().__init__()
        self.out_channels = out_channels
        self.register_buffer("weight", 1.0 / 10 ** torch.linspace(0, 9, out_channels).view(1, -1))

    def forward(self, x):
        return math.sqrt(1.0 / self.out_channels) * torch.cos(x.reshape(-1, 1).to(self.weight.dtype) * self.weight)


class TemporalPooling(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, num_segments: int, encoding: Optional[str] = None,
                 encoding_level: str = "frame"):
        super().__init__()
        self.input_size, self.output_size, self.num_segments = input_size, output_size, num_segments
        self.encoding_level = encoding_level
        self.encoding = self._build_positional_encoding(input_size, encoding, num_segments)
        # MLP applied on top of the positional encoding (pooling.py:44-47)
        self.encoding_mlp = Linear(input_size, input_size) if self.encoding is not None else None

    def _build_positional_encoding(self, input_size: int, encoding: Optional[str], num_segments: int):
        if encoding == "positional":
            return PositionalEncoding(input_size)
        if encoding == "temporal":
            return TemporalEncoding(input_size)
        if encoding == "learnt":
            if self.encoding_level == "frame":
                return torch.nn.Parameter(torch.rand((num_segments, input_size)), requires_grad=True)
            logger.warning("Learnt encoding is supported only for frame level encoding!")
        logger.warning("No positional encoding in use")  # (pooling.py:61: also reached for an unsupported request)
        return None

    def _rows(self, pos: torch.Tensor) -> torch.Tensor:
        """encoding(pos) as [len(pos), input_size] f32 rows on the device."""
        if isinstance(self.encoding, PositionalEncoding):
            zeros = torch.zeros((pos.numel(), self.input_size), dtype=torch.float32, device=pos.device)
            return ops.pe_add(zeros, pos.reshape(-1).long(), self.encoding.frequency).float()
        return self.encoding(pos.float()).float()

    def apply_positional_embedding(self, x: torch.Tensor, batch: torch.Tensor, pos: torch.Tensor):
        """x [N, S, F], batch [N], pos [N] -> x + encoding (pooling.py:64-83)."""
        if self.encoding is None:
            return x
        mlp = lambda rows: ops.linear(ops.to_act(rows), self.encoding_mlp.weight, self.encoding_mlp.bias).to(x.dtype)
        if self.encoding_level == "frame":
            if isinstance(self.encoding, torch.nn.Parameter):
                return x + mlp(self.encoding).unsqueeze(0)
            return x + mlp(self._rows(torch.arange(0, self.num_segments, device=x.device))).unsqueeze(0)
        # action / video level: every row of the batch gets the encoding of its own position (the reference loops over the
        # batch ids and fills a zero tensor: every row belongs to exactly one id, so this is the same tensor)
        return x + mlp(self._rows(pos)).unsqueeze(1)

    def forward(self, x, batch, pos):
        raise NotImplementedError("TemporalPooling.forward is not implemented")
