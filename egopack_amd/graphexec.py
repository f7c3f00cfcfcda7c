"""Segmented replay of a captured step (``egk_graph_plan_*``, csrc/graph_exec.hip).

``torch.cuda.CUDAGraph.replay()`` hands the captured graph to the HIP runtime, which enqueues the nodes of any graph with a
fork one by one (2.8-4.2 us of host time per node, chain after chain in creation order).  ``SegmentedGraph`` replays the same
nodes as single-stream graphs -- one per fork-free path, on the runtime's batch path -- joined by events.  The reference has
no counterpart (it steps eagerly: main_temporal.py:78-104, main_egopack.py:72-99); this is the replay half of the captured
step of engine.StepBase.capture."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class SegmentedGraph:
    """The nodes of ``graph`` (captured with ``torch.cuda.CUDAGraph(keep_graph=True)``) as a plan of single-stream graphs.
    Keeps ``graph`` alive: the nodes address memory of its capture pool."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", max_streams: int = 4, event_nodes: bool = False):
        lib = _lib.load()
        raw = graph.raw_cuda_graph()
        plan = C.c_void_p()
        rc = lib.egk_graph_plan_create(C.c_void_p(int(raw)), int(max_streams), int(bool(event_nodes)), C.byref(plan))
        if rc != 0:
            raise RuntimeError(f"egk_graph_plan_create failed (code {rc}): {_lib.last_error()}")
        self._lib, self._plan, self._graph = lib, plan, graph

    def replay(self) -> None:
        rc = self._lib.egk_graph_plan_launch(self._plan, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"egk_graph_plan_launch failed (code {rc}): {_lib.last_error()}")

    def pool(self):
        return self._graph.pool()

    def info(self) -> dict:
        v = [C.c_int32() for _ in range(4)]
        self._lib.egk_graph_plan_info(self._plan, *[C.byref(x) for x in v])
        return dict(zip(("nodes", "segments", "streams", "cross_edges"), (x.value for x in v)))

    def segments(self):
        """[(nodes, stream, waits, records)] in launch order."""
        out = []
        d = (C.c_int32 * 4)()
        for s in range(self.info()["segments"]):
            self._lib.egk_graph_plan_segment(self._plan, s, d)
            out.append(tuple(d))
        return out

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            self._lib.egk_graph_plan_destroy(plan)
