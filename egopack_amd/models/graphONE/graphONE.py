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
        self._searched: Dict[tuple, tuple] = {}  # results of ``search_ahead`` waiting for their ``interact``
        self._forked = False  # the last ``_search_all`` ran on the task streams

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
        ready = self._searched.pop(tuple(id(f) for f in feats), None)
        nn_idx, f_act = ready[:2] if ready is not None else self._search_all(items)
        self.stream_of = {}  # (every output is made on the caller's stream)
        outs = ops.graphone_stages(f_act, banks, [nn_idx[t] for t in tasks], stage_lists, self.residual)
        output = dict(zip(tasks, outs))
        closest = {t: [nn_idx[t][:, 0]] * self.depth for t in tasks}
        return output, closest

    def _search_all(self, items, join: bool = True):
        """({task: [N, k] prototype indices}, the features in the activation type): the searches side by side on the task streams,
        forked from the current stream, the cast beside them on the current stream; ``join``: the task streams are joined back into
        the current stream (else the caller joins ``ahead_streams()`` where the results are consumed)."""
        main = torch.cuda.current_stream()
        feats = [f for _, f in items]
        nn_idx = {}
        banks = [self.embeddings[t].weight for t, _ in items]
        if ops.act_dtype() == torch.bfloat16 and ops.nearest_prototypes_grouped_ok(feats, banks, self.k, self.distance_func):
            # every task's search as ONE chain of grouped launches on this stream (no fork: ``ahead_streams`` stays empty)
            lists, f_act = ops.nearest_prototypes_grouped(feats, banks, self.k, [self._bank_norm(t) for t, _ in items])
            self._forked = False
            return dict(zip([t for t, _ in items], lists)), f_act
        self._forked = bool(self.parallel_tasks)
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
                if join:
                    main.wait_stream(st)
                    nn_idx[task].record_stream(main)
        else:
            f_act = ops.to_act_rows(feats)
            for task, f in items:
                nn_idx[task] = ops.nearest_prototypes(f.detach(), self.embeddings[task].weight, self.k, self.distance_func,
                                                      self._bank_norm(task))
        return nn_idx, f_act

    def search_ahead(self, features: Dict[str, torch.Tensor]) -> bool:
        """Run the prototype searches of a coming ``interact(features)`` NOW, forked from the current stream -- the searches need
        nothing but the features, while ``interact`` is issued where the stages can run, behind whatever else its caller's stream
        holds: the EgoPack step forks the searches from the stream of the pass that makes the features BEFORE that stream joins the
        training pass's, so they run beside the training pass's forward chain instead of behind it.  The results are kept for the
        ``interact`` call over the SAME tensors.  The CALLER joins ``ahead_streams()`` into the stream that calls ``interact``: the
        task streams are not joined back into the current stream, because under capture this stream may itself be a fork of the
        capture's origin stream, and ROCm 7.2's capture bookkeeping lists the waiting stream as a parallel stream of the event's
        stream on EVERY wait of a non-origin stream -- fork A -> B plus join B -> A makes the two lists a cycle and
        ``hipStreamEndCapture`` recurse until the stack ends.  False: the grouped interaction does not apply (nothing was run)."""
        items = list(features.items())
        feats = [f for _, f in items]
        if (len(items) < 2 or not feats[0].is_cuda or not torch.is_grad_enabled() or any(f.requires_grad for f in feats)
                or any(f.dim() != 2 or f.shape != feats[0].shape for f in feats) or feats[0].shape[0] % 64 or feats[0].shape[1] % 64
                or not ops.graphone_stages_ok(len(items), feats[0].shape[0], feats[0].shape[1], [self.embeddings[t].weight for t, _ in items],
                                              [self.conv_stages[t] for t, _ in items], self.freeze)):
            return False
        if self._searched and self._forked:
            return False  # (one set of task streams: a second pending search would need the first one joined)
        with torch.no_grad():
            nn_idx, f_act = self._search_all(items, join=False)
        self._searched[tuple(id(f) for f in feats)] = (nn_idx, f_act, feats)  # (feats held: their ids stay theirs)
        return True

    def ahead_streams(self):
        """The streams pending ``search_ahead`` results were made on, beside the stream ``search_ahead`` was called on."""
        return list(self._task_streams) if self._searched and self._forked else []

    def drop_searched(self) -> None:
        """Forget pending ``search_ahead`` results (a result nobody fetched must not outlive its step)."""
        self._searched.clear()

    def searched_tensors(self):
        """The tensors of the pending ``search_ahead`` results (for the caller's stream bookkeeping)."""
        out = []
        for nn_idx, f_act, _ in self._searched.values():
            out += list(nn_idx.values()) + list(f_act if isinstance(f_act, (list, tuple)) else [f_act])
        return out

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
