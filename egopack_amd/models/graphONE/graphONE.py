"""Cross-task prototype GNN ``GraphONE``.

Mirror of reference models/graphONE/graphONE.py:13-151: same constructor (extra config keys are
swallowed), ``embeddings`` / ``conv_stages.<task>.<d>.module_{0,1,3}`` state-dict layout,
``interact(dict) -> (dict, dict)``.

What the reference computes per aux task and depth d (graphONE.py:94-115):
    edges  = k nearest prototypes of the ORIGINAL features (cosine), recomputed every depth
    graph  = cat([bank, f]); SAGEConv(max) over graph with self loops; LayerNorm; ReLU; Linear
    f      = graph[-N:] (+ f if residual)
Only the last N rows are kept and the bank is frozen, so the K prototype rows of the stage are
dead work, and the edges are identical at every depth.  Here, per aux task:
    nn = nearest_prototypes(f0, bank)       ONCE  (exact-f32 MFMA product + wave top-k; cosine or cdist/4096)
    per depth:  m = gather_max(f, bank, nn) (max over the k prototype rows and the self row)
                h = [m | f].[Wl | Wr]^T -> LayerNorm+ReLU -> Linear (+ f residual in the epilogue)
which gives the same N output rows.  With ``freeze=False`` the prototypes are parameters: their gradient comes
through the max aggregation only (the bank rows' own stage outputs are discarded by the reference, the search runs
under no_grad) and is a deterministic gather over the edge list grouped by prototype (ops._GatherMax.backward).
"""
from __future__ import annotations

import logging
from typing import Dict, List, Literal, Tuple

import torch
import torch.nn as nn

from ... import ops
from ..layers import LayerNorm, Linear, SAGEConv

logger = logging.getLogger(__name__)


class GraphONE(nn.Module):
    def __init__(self, graphone: Dict[str, torch.Tensor], features_size: int = 1024, hidden_size: int = 1024,
                 freeze: bool = True, k: int = 8, depth: int = 3, distance_func: Literal["l2", "cosine"] = "cosine",
                 residual: bool = False, mix_strategy: Literal["mean", "max", "transformer"] = "max",
                 update_edges_interval: int = 1, share_params: bool = False, *args, **kwargs) -> None:
        super().__init__()
        self.feature_size = features_size
        self.k, self.distance_func, self.residual, self.mix_strategy = k, distance_func, residual, mix_strategy
        self.update_edges_interval, self.share_cnn_params = update_edges_interval, share_params
        self.depth = depth
        logger.info("GraphONE: %d tasks, depth=%d, K=%d", len(graphone), depth, k)
        if not freeze:
            logger.warning("GraphONE initialized with trainable prototypes.")
        self.freeze = freeze
        self.task_labels = sorted(graphone.keys())
        self.embeddings = nn.ModuleDict({t: nn.Embedding.from_pretrained(graphone[t], freeze=freeze)
                                         for t in self.task_labels})
        stages = {}
        for t in self.task_labels:
            per_depth = []
            for _ in range(depth):
                stage = nn.Module()
                stage.module_0 = SAGEConv(features_size, hidden_size, aggr="max", project=False, bias=False)
                stage.module_1 = LayerNorm(hidden_size)
                stage.module_2 = nn.ReLU()
                stage.module_3 = Linear(hidden_size, features_size)
                per_depth.append(stage)
            stages[t] = nn.ModuleList(per_depth)
        self.conv_stages = nn.ModuleDict(stages)
        self._bank_inv_norm: Dict[str, torch.Tensor] = {}
        # MI355X knob (not a reference key): the aux tasks' stages are independent chains of M = batch-nodes
        # contractions that each fill half of the chip at best -- run them on one HIP stream per task
        self.parallel_tasks = bool(kwargs.get("parallel_tasks", True))
        self._task_streams: List[torch.cuda.Stream] = []

    def _bank_norm(self, task: str) -> torch.Tensor:
        """Per-prototype 1/||p|| (cosine) or ||p||^2 (l2): cached while the bank is frozen, recomputed per call otherwise
        (trainable prototypes move with every optimizer step)."""
        bank = self.embeddings[task].weight
        cached = self._bank_inv_norm.get(task)
        if (cached is None or cached.device != bank.device or cached.shape[0] != bank.shape[0]
                or not self.freeze or cached._version_of != bank._version or cached._kind != self.distance_func):
            cached = (ops.row_sq_norm if self.distance_func == "l2" else ops.row_inv_norm)(bank.detach())
            cached._version_of, cached._kind = bank._version, self.distance_func
            self._bank_inv_norm[task] = cached
        return cached

    def interact(self, features: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, List[torch.Tensor]]]:
        if self.distance_func not in ("cosine", "l2"):
            raise ValueError(f"Unknown distance function: {self.distance_func}")  # reference graphONE.py:131
        output, closest = {}, {}
        items = list(features.items())
        grouped = self._grouped_interaction(items)
        if grouped is not None:
            return grouped
        if self.parallel_tasks and len(items) > 1 and items[0][1].is_cuda:
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            while len(self._task_streams) < len(items):
                self._task_streams.append(torch.cuda.Stream())
                ops.exclude_wgrad_streams(self._task_streams[-1:])
            self.stream_of = {task: st for st, (task, _) in zip(self._task_streams, items)}  # (the stream each output was made on)
            for st, (task, f) in zip(self._task_streams, items):
                st.wait_event(fork)
                f.record_stream(st)
                with torch.cuda.stream(st):  # autograd replays each chain's backward on its stream as well
                    output[task], closest[task] = self._task_interaction(task, f)
            for st, (task, _) in zip(self._task_streams, items):
                main.wait_stream(st)
                output[task].record_stream(main)
                for a in closest[task][:1]:
                    a.record_stream(main)
            return output, closest
        for task, f in items:
            output[task], closest[task] = self._task_interaction(task, f)
        return output, closest

    def _grouped_interaction(self, items):
        """The interaction of all tasks as ONE chain of grouped launches (ops.graphone_stages) when it qualifies -- equal row
        counts, bf16 activations, frozen banks, parameters in the optimizer's flat buffers (the EgoPack training step) -- else
        None: G chains over a third of the rows each fill the chip worse than one chain over all of them, and a captured step
        has ~100 launches fewer.  The prototype searches stay one per task, side by side on the task streams."""
        if len(items) < 2 or not items[0][1].is_cuda or not torch.is_grad_enabled():
            return None
        tasks = [t for t, _ in items]
        feats = [f for _, f in items]
        if (any(f.requires_grad for f in feats) or any(f.dim() != 2 or f.shape != feats[0].shape for f in feats)
                or feats[0].shape[0] % 64 or feats[0].shape[1] % 64):
            return None
        banks = [self.embeddings[t].weight for t in tasks]
        stage_lists = [self.conv_stages[t] for t in tasks]
        if not ops.graphone_stages_ok(len(items), feats[0].shape[0], feats[0].shape[1], banks, stage_lists, self.freeze):
            return None
        main = torch.cuda.current_stream()
        nn_idx = {}
        if self.parallel_tasks:
            fork = torch.cuda.Event()
            fork.record(main)
            while len(self._task_streams) < len(items):
                self._task_streams.append(torch.cuda.Stream())
                ops.exclude_wgrad_streams(self._task_streams[-1:])
            for st, (task, f) in zip(self._task_streams, items):
                st.wait_event(fork)
                f.record_stream(st)
                with torch.cuda.stream(st):
                    nn_idx[task] = ops.nearest_prototypes(f.detach(), self.embeddings[task].weight, self.k, self.distance_func,
                                                          self._bank_norm(task))
            f_act = ops.to_act_rows(feats)  # (on the caller's stream, beside the searches)
            for st, (task, _) in zip(self._task_streams, items):
                main.wait_stream(st)
                nn_idx[task].record_stream(main)
        else:
            f_act = ops.to_act_rows(feats)
            for task, f in items:
                nn_idx[task] = ops.nearest_prototypes(f.detach(), self.embeddings[task].weight, self.k, self.distance_func,
                                                      self._bank_norm(task))
        self.stream_of = {}  # (every output is made on the caller's stream)
        outs = ops.graphone_stages(f_act, banks, [nn_idx[t] for t in tasks], stage_lists, self.residual)
        output = dict(zip(tasks, outs))
        closest = {t: [nn_idx[t][:, 0]] * self.depth for t in tasks}
        return output, closest

    def _task_interaction(self, task: str, features: torch.Tensor):
        bank = self.embeddings[task].weight
        # the search reads the features as they are handed in: callers that want index selection independent of the
        # activation storage type pass the f32 output of the producing contraction (engine.EgoPackStep does)
        # (the bank is handed over as the parameter itself: its bf16 halves for the search product are cached on it)
        nn_idx = ops.nearest_prototypes(features.detach(), bank, self.k, self.distance_func, self._bank_norm(task))
        assignments = [nn_idx[:, 0]] * self.depth  # the reference recomputes identical edges per depth
        f = ops.to_act(features)
        for stage in self.conv_stages[task]:
            m = ops.gather_max(f, bank, nn_idx)
            h = stage.module_0.combine(m, f)
            h = stage.module_1(h, relu=True)
            f = stage.module_3(h, residual=f if self.residual else None)
        return f, assignments


def cos_dissimilarity(g1: torch.Tensor, g2: torch.Tensor) -> torch.Tensor:
    """1 - cosine similarity [N, K] (reference graphONE.py:148-151) on the exact-f32 MFMA path."""
    inv1, inv2 = ops.row_inv_norm(g1), ops.row_inv_norm(g2)
    dot = torch.empty((g1.shape[0], g2.shape[0]), dtype=torch.float32, device=g1.device)
    ops.gemm(g1.shape[0], g2.shape[0], g1.contiguous(), g1.shape[1], g2.contiguous(), g2.shape[1], g1.shape[1], dot,
             g2.shape[0], compute=ops.F32)
    return ops.scaled_one_minus(dot, inv1, inv2)
