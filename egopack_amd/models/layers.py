"""Parameter holders with torch's default initialisation whose forward launches the HIP kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class Linear(nn.Linear):
    """nn.Linear / gnn.Linear parameters (kaiming-uniform(a=sqrt 5), bias U(+-1/sqrt(fan_in)));
    forward = one MFMA launch with the bias (and optional ReLU / residual) in the epilogue."""

    def forward(self, x, residual=None, relu=False, out_f32=False, ln_in=None, res_sink=None, slab_ok=False):
        return ops.linear(x, self.weight, self.bias, residual=residual, relu=relu, out_f32=out_f32, ln_in=ln_in, res_sink=res_sink,
                          slab_ok=slab_ok)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters; forward = fused LayerNorm(+ReLU)(+dropout) launch."""

    def forward(self, x, relu=False, p=0.0):
        return ops.row_layernorm(x, self.weight, self.bias, self.eps, relu=relu, p=p, training=self.training)


class Dropout(nn.Dropout):
    def forward(self, x):
        return ops.dropout(x, self.p, self.training)


class GraphLayerNorm(nn.Module):
    """gnn.LayerNorm(C, mode='graph') parameters (weight=1, bias=0)."""

    def __init__(self, in_channels: int, eps: float = 1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))

    def forward(self, x, seg_ptr, slope=0.2, **kw):
        return ops.graph_layernorm_lrelu(x, self.weight, self.bias, seg_ptr, self.eps, slope, **kw)


class PositionalEncoding(nn.Module):
    """gnn.PositionalEncoding(C): buffer ``frequency`` = logspace(0, 1, C/2, base 1e-4)."""

    def __init__(self, out_channels: int, base_freq: float = 1e-4):
        super().__init__()
        if out_channels % 2 != 0:
            raise ValueError(f"Cannot use sinusoidal positional encoding with odd 'out_channels' (got {out_channels}).")
        self.out_channels = out_channels
        self.register_buffer("frequency", torch.logspace(0, 1, out_channels // 2, base_freq))

    def add_to(self, x, pos, pos_range=None):
        return ops.pe_add(x, pos, self.frequency, pos_range)


class SAGEConv(nn.Module):
    """Parameters of gnn.SAGEConv in PyG's registration order (lin, lin_l, lin_r)."""

    def __init__(self, in_channels: int, out_channels: int, aggr: str = "mean", project: bool = False, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.aggr, self.project = in_channels, out_channels, aggr, project
        if project:
            self.lin = Linear(in_channels, in_channels, bias=True)
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.lin_r = Linear(in_channels, out_channels, bias=False)

    def combine(self, agg, x):
        """lin_l(agg) + lin_r(x) as ONE two-source contraction."""
        return ops.linear(agg, self.lin_l.weight, self.lin_l.bias, x2=x, W2=self.lin_r.weight)
