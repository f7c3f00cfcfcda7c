"""Multi-scale temporal relation module (reference models/TRN.py:9-74: ``RelationModuleMultiScale``).

The reference keeps this class but nothing imports it; it is built here as the one module whose
parity is pinned by the reference class ITSELF (tests/golden/trn_multiscale.pt, oracle/make_golden_trn_multiscale.py:
it imports with no stand-ins).

Reference semantics: for every scale s in (num_frames .. 2), out[:, scale] = sum over the selected s-frame relations r
(all C(F, s) combinations, the first one for the largest scale, up to 3 evenly spaced ones for the others) of
``ReLU(Linear_s(ReLU(concat of the frames of r))))``.

Here: ReLU is applied to the [B, F, D] input ONCE (it commutes with selecting frames), and every scale is ONE MFMA
contraction: the rows of its selected relations are gathered side by side -- row (b, r) of a [B * n_r, s * D] operand -- so
the scale's n_r relations share one launch of ``ops.linear`` (bias + ReLU in the epilogue; backward: dX, and dW / db straight
into the optimizer's flat gradient buffer).  The n_r row groups are then summed.  Constructor, attributes and state-dict
keys (``fc_fusion_scales.{i}.1.{weight,bias}``) are the reference's."""
from __future__ import annotations

import itertools
import logging
from math import ceil

import torch
import torch.nn as nn

from .. import ops
from .layers import Linear

logger = logging.getLogger(__name__)


class RelationModuleMultiScale(nn.Module):
    def __init__(self, img_feature_dim, num_bottleneck, num_frames, verbose=False):
        super().__init__()
        self.subsample_num = 3  # relations summed per scale (TRN.py:14)
        self.img_feature_dim = img_feature_dim
        self.num_bottleneck = num_bottleneck
        self.num_frames = num_frames
        self.scales = list(range(num_frames, 1, -1))  # TRN.py:16
        self.relations_scales = [self.return_relationset(num_frames, s) for s in self.scales]
        self.subsample_scales = [min(self.subsample_num, len(r)) for r in self.relations_scales]
        # containers with the reference's key layout; forward below never calls them
        self.fc_fusion_scales = nn.ModuleList(
            nn.Sequential(nn.ReLU(), Linear(s * img_feature_dim, num_bottleneck), nn.ReLU()) for s in self.scales)
        if verbose:
            logger.debug("Multi-Scale Temporal Relation Network Module in use: %s", ["%d-frame relation" % i for i in self.scales])
        self._index_cache = {}

    def return_relationset(self, num_frames, num_frames_relation):
        return list(itertools.combinations(range(num_frames), num_frames_relation))  # TRN.py:71-74

    def selected_relations(self, scale_id: int):
        """The relations a forward pass evaluates for one scale (TRN.py:45 for the largest, :56-60 evenly spaced otherwise)."""
        rel = self.relations_scales[scale_id]
        if scale_id == 0:
            return [rel[0]]
        n_total, n_sel = len(rel), self.subsample_scales[scale_id]
        return [rel[int(ceil(i * n_total / n_sel))] for i in range(n_sel)]

    def _frame_index(self, scale_id: int, device):
        key = (scale_id, str(device))
        idx = self._index_cache.get(key)
        if idx is None:
            flat = [f for r in self.selected_relations(scale_id) for f in r]
            idx = self._index_cache[key] = torch.tensor(flat, dtype=torch.long, device=device)
        return idx

    def forward(self, input):
        if input.dim() != 3 or input.shape[1] != self.num_frames or input.shape[2] != self.img_feature_dim:
            raise ValueError(f"expected [B, {self.num_frames}, {self.img_feature_dim}], got {tuple(input.shape)}")
        B = input.shape[0]
        xr = ops.relu(ops.to_act(input))  # the first ReLU of every fc_fusion, once
        outs = []
        for sid, s in enumerate(self.scales):
            n_r = len(self.selected_relations(sid))
            rows = xr.index_select(1, self._frame_index(sid, xr.device)).view(B * n_r, s * self.img_feature_dim)
            lin = self.fc_fusion_scales[sid][1]
            y = ops.linear(rows, lin.weight, lin.bias, relu=True).view(B, n_r, self.num_bottleneck)
            outs.append(y[:, 0] if n_r == 1 else y.sum(dim=1))
        return torch.stack(outs, dim=1)  # [B, scales, bottleneck], largest scale first (TRN.py:49-69)
