"""Oracle restatement of the torch_geometric==2.3.0 leaf ops used on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for this file: the
source of torch_geometric 2.3.0 / torch_cluster 1.6.1 / torch_scatter 2.1.1 (reference
environment.yml:174,184,188) is not under /root/reference and is not installed, and the
reference has no tests at this boundary.  Each function restates the published algorithm
of that release and names the reference call site that relies on it; hand-derived
known-answer tests live in tests/test_oracle_known_answers.py.

All functions are pure torch (any device, used on CPU), differentiable where the reference
differentiates through them.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import storage as S


# --------------------------------------------------------------------------------------
# scatter / aggregation primitives
# --------------------------------------------------------------------------------------
def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """torch_geometric.utils.scatter(src, index, dim=0, dim_size, reduce='sum')
    = ``src.new_zeros(size).scatter_add_``.  Call site: reference graphone.py:53."""
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return out.scatter_add_(0, idx, src)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """MeanAggregation: scatter_add / clamp(count, min=1); rows without entries are 0.
    Call site: SAGEConv(aggr='mean') at reference models/graph.py:42."""
    s = scatter_sum(src, index, dim_size)
    cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
    cnt.scatter_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    return s / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))


class _ScatterMaxFirst(torch.autograd.Function):
    """amax per index group whose backward sends the whole gradient to ONE winner, the first source row (in source order) that
    attains the maximum -- the rule of the product's max kernels and of torch_scatter's arg-max on CUDA.  Used only
    under the bf16 storage model (oracle/storage.py): among bf16 values exact ties are common (8 significant bits), and
    ``scatter_reduce('amax')`` would split a tied gradient evenly where the product routes it to one row."""

    @staticmethod
    def forward(ctx, src, index, dim_size):
        out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        out.scatter_reduce_(0, idx, src, reduce="amax", include_self=False)
        pos = torch.arange(src.shape[0], device=src.device).view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        big = src.shape[0]
        cand = torch.where(src == out.gather(0, idx), pos, torch.full_like(pos, big))
        first = torch.full(out.shape, big, dtype=pos.dtype, device=src.device).scatter_reduce_(0, idx, cand, reduce="amin", include_self=True)
        ctx.save_for_backward(idx, pos == first.gather(0, idx))
        return out

    @staticmethod
    def backward(ctx, g):
        idx, win = ctx.saved_tensors
        return g.gather(0, idx) * win.to(g.dtype), None, None


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """MaxAggregation on the CPU / no-torch_scatter branch of PyG 2.3.0:
    ``new_zeros(size).scatter_reduce_(0, index, src, 'amax', include_self=False)``;
    rows without entries stay 0.  Call sites: SAGEConv(aggr='max') reference
    models/graphONE/graphONE.py:60 and global_max_pool reference models/tasks/oscc.py:68,85.
    Backward on exact ties splits evenly here (torch_scatter picks one winner): SURVEY A.1."""
    if S.is_on():  # (the product's single-winner tie rule: see _ScatterMaxFirst)
        return _ScatterMaxFirst.apply(src, index, dim_size)
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return out.scatter_reduce_(0, idx, src, reduce="amax", include_self=False)


def global_max_pool(x: torch.Tensor, batch: torch.Tensor, size: Optional[int] = None) -> torch.Tensor:
    """gnn.pool.global_max_pool: scatter(x, batch, dim=-2, dim_size=int(batch.max())+1, 'max')."""
    if size is None:
        size = int(batch.max()) + 1
    return scatter_max(x, batch, size)


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gnn.Linear / nn.Linear forward: x @ W.T + b."""
    return F.linear(x, weight, bias)


def sage_conv(
    x: torch.Tensor,
    edge_index: torch.Tensor,
    lin_l_weight: torch.Tensor,
    lin_l_bias: Optional[torch.Tensor],
    lin_r_weight: torch.Tensor,
    lin_weight: Optional[torch.Tensor] = None,
    lin_bias: Optional[torch.Tensor] = None,
    aggr: str = "mean",
) -> torch.Tensor:
    """SAGEConv(in, out, aggr, root_weight=True, project=lin_weight is not None).

    x_src = relu(lin(x)) if project else x; messages x_src[edge_index[0]] are reduced at
    edge_index[1] (flow source->target); out = lin_l(agg) + lin_r(x) with the UN-projected x.
    Call sites: reference models/graph.py:42 (mean, project=True, bias=True) and
    models/graphONE/graphONE.py:60 (max, project=False, bias=False)."""
    n = x.shape[0]
    # (S.act / S.weight: identity unless the bf16 storage model of oracle/storage.py is on -- the projected features, the
    #  aggregate and the layer's output are the three tensors the product stores)
    x_src = S.act(F.relu(F.linear(x, S.weight(lin_weight), lin_bias))) if lin_weight is not None else x
    msg = x_src.index_select(0, edge_index[0])
    if aggr == "mean":
        agg = scatter_mean(msg, edge_index[1], n)
    elif aggr == "max":
        agg = scatter_max(msg, edge_index[1], n)
    else:
        raise ValueError(aggr)
    agg = S.act(agg)
    return F.linear(agg, S.weight(lin_l_weight), lin_l_bias) + F.linear(x, S.weight(lin_r_weight))


def graph_layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """gnn.LayerNorm(C, mode='graph') called with batch=None (reference models/graph.py:43:
    the Sequential routes "x -> x", so no batch vector ever reaches it):
    x = x - x.mean(); out = x / (x.std(unbiased=False) + eps); out * weight + bias.
    Statistics span ALL N*C elements; eps is added to the std, not the variance."""
    x = x - x.mean()
    out = x / (x.std(unbiased=False) + eps)
    return out * weight + bias


def positional_encoding_frequency(channels: int, base_freq: float = 1e-4) -> torch.Tensor:
    """Buffer of gnn.PositionalEncoding: torch.logspace(0, 1, C//2, base_freq)."""
    return torch.logspace(0, 1, channels // 2, base_freq)


def positional_encoding(pos: torch.Tensor, frequency: torch.Tensor) -> torch.Tensor:
    """gnn.PositionalEncoding.forward (granularity 1): out = pos.view(-1,1)*freq.view(1,-1);
    cat([sin(out), cos(out)], -1).  Call site: reference models/graph.py:37,63."""
    out = pos.view(-1, 1) * frequency.view(1, -1)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=-1)


# --------------------------------------------------------------------------------------
# edge utilities (integer ops: bit-exact)
# --------------------------------------------------------------------------------------
def add_remaining_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.add_remaining_self_loops without attributes: drop existing self
    loops, append [arange(N); arange(N)].  Call site: reference graphONE.py:109."""
    mask = edge_index[0] != edge_index[1]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index[:, mask], loops.unsqueeze(0).repeat(2, 1)], dim=1)


def coalesce(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """RemoveDuplicatedEdges = coalesce: sort by row*N+col and drop duplicates.
    Call site: reference models/transforms/lta_temp_connectivity.py:28,56."""
    key = edge_index[0] * num_nodes + edge_index[1]
    key = torch.unique(key, sorted=True)
    return torch.stack([key // num_nodes, key % num_nodes])


def radius_graph(
    pos: torch.Tensor,
    r: float,
    batch: Optional[torch.Tensor] = None,
    loop: bool = False,
    max_num_neighbors: int = 32,
) -> torch.Tensor:
    """torch_cluster.radius_graph(flow='source_to_target') on 1-D / [N,1] positions:
    all (source j, target i) with ||pos_i - pos_j|| < r (strict, as torch_cluster's radius test: SURVEY A.9) inside one batch element, grouped
    by target, at most max_num_neighbors per target (never binding for the band graphs the
    reference builds: r = k + 0.5, k <= 2).  Neighbour order inside a target group is
    KD-tree dependent in torch_cluster; here it is ascending source index.
    Call sites: RadiusGraph(r=cfg.k+0.5) reference main_temporal.py:168,189,225 and
    lta_temp_connectivity.py:37-45."""
    p = pos.reshape(pos.shape[0], -1).to(torch.float64)
    n = p.shape[0]
    d = torch.cdist(p, p)
    adj = d < r
    if batch is not None:
        adj &= batch.view(-1, 1) == batch.view(1, -1)
    if not loop:
        adj &= ~torch.eye(n, dtype=torch.bool, device=pos.device)
    tgt, src = adj.nonzero(as_tuple=True)  # row-major: grouped by target, ascending source
    if max_num_neighbors is not None:
        deg = torch.bincount(tgt, minlength=n)
        start = torch.cumsum(deg, 0) - deg
        rank = torch.arange(tgt.numel(), device=pos.device) - start[tgt]
        keep = rank < max_num_neighbors
        tgt, src = tgt[keep], src[keep]
    return torch.stack([src, tgt])


# --------------------------------------------------------------------------------------
# batching
# --------------------------------------------------------------------------------------
class OData:
    """Minimal stand-in for torch_geometric.data.Data / Batch: attribute bag."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __contains__(self, key):
        return hasattr(self, key) and getattr(self, key) is not None

    def to(self, device, non_blocking: bool = False):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self


def collate(samples: Sequence[OData]) -> OData:
    """PyG Batch.from_data_list: concat x / y / pos along dim 0, offset-and-concat edge_index
    along dim -1, add ``batch`` and ``ptr``; python numbers become a [B] tensor.
    Call site: reference utils/dataloading.py:56-70."""
    xs, ys, poss, eis, batch, ptr = [], [], [], [], [], [0]
    off = 0
    for b, s in enumerate(samples):
        n = s.x.shape[0]
        xs.append(s.x)
        poss.append(s.pos)
        if torch.is_tensor(s.y):
            ys.append(s.y)
        else:
            ys.append(torch.tensor([s.y]))
        eis.append(s.edge_index + off)
        batch.append(torch.full((n,), b, dtype=torch.long))
        off += n
        ptr.append(off)
    return OData(
        x=torch.cat(xs),
        y=torch.cat(ys),
        pos=torch.cat(poss),
        edge_index=torch.cat(eis, dim=1),
        batch=torch.cat(batch),
        ptr=torch.tensor(ptr, dtype=torch.long),
        num_graphs=len(samples),
    )
