"""Temporal backbone ``Graph``: TRN pooling -> depth x (SAGEConv-mean -> graph LayerNorm ->
LeakyReLU) -> Linear, with a residual.

Mirror of reference models/graph.py:15-65: same constructor, ``forward(data)`` reading
``x, pos, edge_index, batch``, ``configure_optimizers``, and PyG's ``net.module_<i>`` state-dict
keys.  MI355X-first differences (results identical up to fp summation order):
  * SAGE ``lin_l(agg) + lin_r(x)`` is one two-source MFMA contraction; bias / ReLU / residual live
    in contraction epilogues; LayerNorm+LeakyReLU and the neighbour mean are single launches;
  * ``forward`` accepts a MERGED batch (egopack_amd.data.merge_batches) holding the node sets of
    several task batches: GEMMs then run at M = sum of nodes while the graph-LayerNorm statistics
    stay per task batch (``data.seg_ptr``), which is exactly what the reference computes with one
    backbone call per task batch (main_temporal.py:87-90).
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

from .. import ops, switches
from ..data import build_csr
from .layers import Dropout, GraphLayerNorm, Linear, PositionalEncoding, SAGEConv

logger = logging.getLogger(__name__)


def _instantiate(cfg, *args, **kwargs):
    try:  # the real hydra when the caller's environment has it
        from hydra.utils import instantiate
    except ImportError:
        from ..config import instantiate
    return instantiate(cfg, *args, **kwargs)


class Graph(torch.nn.Module):
    def __init__(self, input_size: int, hidden_size: int = 1024, depth: int = 3, pre_dropout: float = 0,
                 temporal_pooling=None, num_segments: int = 8, *args, **kwargs):
        super().__init__()
        self.num_segments, self.hidden_size, self.depth = num_segments, hidden_size, depth
        self.pre_dropout = Dropout(pre_dropout)
        self.temporal_pooling = (_instantiate(temporal_pooling, input_size, hidden_size, num_segments)
                                 if temporal_pooling else None)
        self.positional_encoding = PositionalEncoding(hidden_size)
        if depth > 0:
            self.net = nn.Module()  # children named as PyG's gnn.Sequential names them
            for d in range(depth):
                setattr(self.net, f"module_{3 * d}", SAGEConv(hidden_size, hidden_size, project=True))
                setattr(self.net, f"module_{3 * d + 1}", GraphLayerNorm(hidden_size))
                setattr(self.net, f"module_{3 * d + 2}", nn.LeakyReLU(negative_slope=0.2))
            setattr(self.net, f"module_{3 * depth}", Linear(hidden_size, hidden_size))

    def configure_optimizers(self, _):
        return self.parameters()

    @staticmethod
    def _graph_of(data):
        g = getattr(data, "graph", None)
        if g is None:  # plain PyG-style batch: derive both CSR orientations from edge_index
            g = build_csr(data.edge_index, data.pos.shape[0])
            if g.rowptr.device != data.pos.device:
                g = g.to(data.pos.device)
            try:
                data.graph = g
            except Exception:
                pass
        return g

    def forward(self, data, *args, **kwargs):
        """``data``: one batch (reference contract) or an egopack_amd.data merged batch whose ``x`` is a
        list of per-task feature blocks (``merge_meta``): the fused multi-task pass."""
        x = data.x
        x = [self.pre_dropout(b) for b in x] if isinstance(x, (list, tuple)) else self.pre_dropout(x)
        if self.temporal_pooling is not None:
            x = self.temporal_pooling(x, getattr(data, "batch", None), data.pos)
        elif not isinstance(x, (list, tuple)):
            x = ops.to_act(x)
        if not hasattr(self, "net"):
            return x
        cut = getattr(self, "stage_cut", None)
        if cut is not None and torch.is_grad_enabled():
            # engine-installed autograd cut at the TRN output (engine.StepBase staged backward): everything below runs
            # on a detached leaf, so the backbone's backward can be issued in two pieces (SAGE stack, then TRN)
            x = cut(x)
        ops.stamp("fwd_trn_done")
        graph = self._graph_of(data)
        seg_ptr = getattr(data, "seg_ptr", None)
        if seg_ptr is None:  # plain batch: one segment; cached on the batch (a host->device copy cannot be captured)
            seg_ptr = torch.tensor([0, x.shape[0]], dtype=torch.int32, device=x.device)
            try:
                data.seg_ptr = seg_ptr
            except Exception:
                pass
        pr = getattr(data, "pos_range", None)  # (min, max) of the positions, known on the host for collated batches
        import os
        if not switches.enabled("pe_table"):
            pr = None
        h = self.positional_encoding.add_to(x, data.pos, tuple(pr) if pr is not None else None)
        # the graph LayerNorm's per-segment sums ride on the epilogue of the contraction that produces its input (forward:
        # the SAGE layer's last contraction; backward: the dX contraction of whatever consumes its output) when the shortest
        # row segment is known on the host (merged / collated batches carry it) -- no statistics pass over the tensor
        min_rows = int(getattr(data, "min_seg_rows", 0) or (x.shape[0] if seg_ptr.numel() == 2 else 0))
        n_seg = seg_ptr.numel() - 1
        ln_prev = None
        # x feeds the stack (through x + PE(pos), gradient = identity) and the final residual: instead of autograd adding
        # the two gradients of x in a pass of its own, the final Linear hands the residual's gradient to the FIRST SAGE layer,
        # which adds it in the epilogue of its dX contraction (both are gradients of the same [N, H] tensor)
        res = {} if (self.depth > 0 and x.requires_grad and h.dtype == x.dtype) else None
        for d in range(self.depth):
            conv = getattr(self.net, f"module_{3 * d}")
            norm = getattr(self.net, f"module_{3 * d + 1}")
            slope = getattr(self.net, f"module_{3 * d + 2}").negative_slope
            req = ({"seg_ptr": seg_ptr, "n_seg": n_seg, "min_rows": min_rows}
                   if min_rows > 0 and not ops.graph_ln_exchange_on() else None)
            c = ops.sage_mean_layer(h, conv, graph, ln_out=req, ln_in=ln_prev, res_src=res if d == 0 else None)
            h, ln_prev = norm(c, seg_ptr, slope, partials=req.get("partials") if req else None, min_seg_rows=min_rows,
                              return_ctx=True)                # SAGEConv -> graph-LN -> LeakyReLU
            ops.stamp("fwd_sage", seq=True)
        last = getattr(self.net, f"module_{3 * self.depth}")
        return last(h, residual=x, ln_in=ln_prev, res_sink=res)  # x + Linear(h): residual in the epilogue
