#!/usr/bin/env python3
"""Do event-record / event-wait NODES order two single-stream graphs within ONE replay (and not against the previous replay's
record)?  main: long chain, stamp a -> side: stamp b, long chain, stamp c -> main: stamp d.  a < b and c < d in every replay."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import ops
from egopack_amd.graphexec import SegmentedGraph

dev = "cuda"
x, y = torch.ones(16_000_000, device=dev), torch.ones(16_000_000, device=dev)
ops.stamps_enable(dev, slots=64)
cap, side = torch.cuda.Stream(), torch.cuda.Stream()


def body():
    main = torch.cuda.current_stream()
    ops.stamp("start")
    for _ in range(20):
        x.mul_(1.0001)
    ops.stamp("a")
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.stamp("b")
        for _ in range(20):
            y.mul_(1.0001)
        ops.stamp("c")
    main.wait_stream(side)
    ops.stamp("d")


with torch.cuda.stream(cap):
    body()
torch.cuda.synchronize()
for mode in (False, True):
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        body()
    sg = SegmentedGraph(g, max_streams=4, event_nodes=mode)
    print("event nodes" if mode else "stream events", sg.info(), sg.segments())
    bad = 0
    for it in range(30):
        with torch.cuda.stream(cap):
            sg.replay()
        torch.cuda.synchronize()
        st = dict(ops.stamps_read())
        ok = st["a"] < st["b"] and st["c"] < st["d"]
        bad += not ok
        if it < 2 or not ok:
            print("  ", {k: round(v, 1) for k, v in st.items()}, "ok" if ok else "ORDER VIOLATED")
    # back to back (no host synchronisation between replays: the previous replay's records are still in flight)
    with torch.cuda.stream(cap):
        for _ in range(10):
            sg.replay()
    torch.cuda.synchronize()
    st = dict(ops.stamps_read())
    print("   back to back:", {k: round(v, 1) for k, v in st.items()}, "ok" if st["a"] < st["b"] and st["c"] < st["d"] else "ORDER VIOLATED",
          "| violations in 30 single replays:", bad)
    del sg, g
