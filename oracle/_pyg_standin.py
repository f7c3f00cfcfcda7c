"""Stand-in ``torch_geometric`` / ``hydra`` / ``torchvision`` modules for oracle/make_golden.py.

TEST INFRASTRUCTURE ONLY; used in the BUILD CONTAINER ONLY (never on the GPU box, never by
the product).  The reference's own modules (/root/reference/models/**, criterion/, graphone.py,
main_temporal.train, main_egopack.train) cannot be imported without torch_geometric 2.3.0 and
hydra, which are absent and not installable here.  This file registers module objects under
those names whose leaf classes forward to the restated ops in oracle/pyg_ops.py, so that the
reference's OWN control flow (residuals, cat([bank, f]), self-loop handling, graph[-N:], edge
refresh rule, detach, logit fusion, loss composition, train loops) executes unmodified and
produces the golden vectors.  Parameter registration order and names follow PyG 2.3.0
(SAGEConv: lin, lin_l, lin_r; Sequential children ``module_<i>``) so state dicts have the
reference checkpoint layout.
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Callable, List, Tuple, Union

import torch
import torch.nn as nn

from . import pyg_ops as P


class Linear(nn.Linear):
    """gnn.Linear: same forward and default init distribution as nn.Linear."""


class SAGEConv(nn.Module):
    def __init__(self, in_channels, out_channels, aggr="mean", normalize=False, root_weight=True,
                 project=False, bias=True, **kw):
        super().__init__()
        assert not normalize and root_weight
        self.aggr = aggr
        self.project = project
        if project:
            self.lin = Linear(in_channels, in_channels, bias=True)
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.lin_r = Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index):
        return P.sage_conv(x, edge_index, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight,
                           self.lin.weight if self.project else None,
                           self.lin.bias if self.project else None, aggr=self.aggr)


class LayerNorm(nn.Module):
    def __init__(self, in_channels, eps=1e-5, affine=True, mode="graph"):
        super().__init__()
        assert affine and mode == "graph"
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))

    def forward(self, x, batch=None):
        assert batch is None  # the reference never passes a batch vector (models/graph.py:43)
        return P.graph_layer_norm(x, self.weight, self.bias, self.eps)


class PositionalEncoding(nn.Module):
    def __init__(self, out_channels, base_freq=1e-4, granularity=1.0):
        super().__init__()
        assert granularity == 1.0
        self.register_buffer("frequency", P.positional_encoding_frequency(out_channels, base_freq))

    def forward(self, x):
        return P.positional_encoding(x, self.frequency)


class TemporalEncoding(nn.Module):  # imported by name only (models/temporal_pooling/pooling.py:2)
    pass


class _Sequential(nn.Module):
    def __init__(self, input_args: str, modules: List[Union[Tuple[Callable, str], Callable]]):
        super().__init__()
        self._inputs = [a.strip() for a in input_args.split(",")]
        self._calls = []
        for i, m in enumerate(modules):
            if isinstance(m, (tuple, list)):
                mod, desc = m
                ins, outs = desc.split("->")
                ins = [a.strip() for a in ins.split(",")]
                outs = [a.strip() for a in outs.split(",")]
            else:
                mod, ins, outs = m, None, None
            setattr(self, f"module_{i}", mod)
            self._calls.append((f"module_{i}", ins, outs))

    def forward(self, *args):
        env = dict(zip(self._inputs, args))
        last = None
        for name, ins, outs in self._calls:
            mod = getattr(self, name)
            if ins is None:
                last = mod(last)
            else:
                last = mod(*[env[a] for a in ins])
                if len(outs) == 1:
                    env[outs[0]] = last
                else:
                    env.update(dict(zip(outs, last)))
        return last


def Sequential(input_args, modules):
    return _Sequential(input_args, modules)


class Data(P.OData):
    @property
    def edge_stores(self):
        return [self]


class BaseTransform:
    def __call__(self, data):
        raise NotImplementedError


class RemoveDuplicatedEdges(BaseTransform):
    def __call__(self, data):
        data.edge_index = P.coalesce(data.edge_index, data.pos.shape[0])
        return data


def _radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
    assert flow == "source_to_target"
    return P.radius_graph(x, r, batch, loop, max_num_neighbors)


def _scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    assert dim == 0 and reduce == "sum"
    return P.scatter_sum(src, index, dim_size)


def _add_remaining_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    return P.add_remaining_self_loops(edge_index, num_nodes), edge_attr


def _instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate for a flat {_target_: 'pkg.mod.Class', **kw} mapping."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    kwargs.pop("_recursive_", None)
    mod, _, name = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), name)
    return cls(*args, **{**cfg, **kwargs})


def install() -> None:
    """Register the stand-ins in sys.modules (build container only)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    pool = mod("torch_geometric.nn.pool", global_max_pool=P.global_max_pool)
    gnn = mod("torch_geometric.nn", SAGEConv=SAGEConv, LayerNorm=LayerNorm, Linear=Linear,
              PositionalEncoding=PositionalEncoding, TemporalEncoding=TemporalEncoding,
              Sequential=Sequential, pool=pool, radius_graph=_radius_graph,
              global_max_pool=P.global_max_pool)
    data = mod("torch_geometric.data", Data=Data)
    utils = mod("torch_geometric.utils", scatter=_scatter, add_remaining_self_loops=_add_remaining_self_loops)
    rde = mod("torch_geometric.transforms.remove_duplicated_edges", RemoveDuplicatedEdges=RemoveDuplicatedEdges)
    tr = mod("torch_geometric.transforms", BaseTransform=BaseTransform, remove_duplicated_edges=rde,
             RemoveDuplicatedEdges=RemoveDuplicatedEdges)
    loader_dl = mod("torch_geometric.loader.dataloader", DataLoader=object)
    loader = mod("torch_geometric.loader", DataLoader=object, dataloader=loader_dl)
    mod("torch_geometric", nn=gnn, data=data, utils=utils, transforms=tr, loader=loader)

    hutils = mod("hydra.utils", instantiate=_instantiate)
    mod("hydra", utils=hutils, main=lambda *a, **k: (lambda f: f))
    mod("omegaconf", OmegaConf=object)
    mod("torchvision")
    mod("wandb")
