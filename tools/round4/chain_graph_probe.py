#!/usr/bin/env python3
"""Host cost and device pacing of a replayed hipGraph by SHAPE: (1) one chain on one stream (the runtime's single-list path),
(2) the same chain with one forked side launch (multi-list path), (3) two chains as ONE graph, (4) the two chains as TWO
single-stream graphs launched on two streams and joined by events.  Every link = clock stamp + elementwise launch.
Usage: python3 tools/round4/chain_graph_probe.py [--n 100] [--elems 400000]"""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--elems", type=int, default=400000)
args = ap.parse_args()
dev = "cuda"
n = args.n
xa, xb, x0 = torch.ones(args.elems, device=dev), torch.ones(args.elems, device=dev), torch.ones(1024, device=dev)
ops.stamps_enable(dev, slots=1024)
s1, s2, cap = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def chain(name, x, k=n):
    for i in range(k):
        ops.stamp(f"{name}{i}")
        x.mul_(1.0001)
    ops.stamp(f"{name}end")


def capture(fn, stream):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
        fn()
    return g


def timed(launch, label, names):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = dict(ops.stamps_read())
    base = min(st[f"{c}0"] for c in names)
    msg = "; ".join(f"{c}: {st[c + '0'] - base:.1f} -> {st[c + 'end'] - base:.1f} us ({(st[c + 'end'] - st[c + '0']) / n:.2f} us / link of 2 launches)" for c in names)
    print(f"{label}: host launch {(t1 - t0) * 1e6:.0f} us, done {(t2 - t0) * 1e6:.0f} us; {msg}")


def one_chain():
    chain("A", xa)


def chain_with_fork():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    with torch.cuda.stream(s1):
        x0.mul_(1.0001)
    chain("A", xa)
    main.wait_stream(s1)


def two_chains():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    with torch.cuda.stream(s1):
        chain("B", xb)
    chain("A", xa)
    main.wait_stream(s1)


g1 = capture(one_chain, cap)
timed(g1.replay, "one chain, one stream       ", "A")
g2 = capture(chain_with_fork, cap)
timed(g2.replay, "one chain + 1 forked launch ", "A")
g3 = capture(two_chains, cap)
timed(g3.replay, "two chains, ONE graph       ", "AB")
ga = capture(lambda: chain("A", xa), s1)
gb = capture(lambda: chain("B", xb), s2)
ev = torch.cuda.Event()


def two_graphs():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()
    main.wait_stream(s1)
    main.wait_stream(s2)


# torch replays a graph on the CURRENT stream
timed(two_graphs, "two chains, TWO graphs      ", "AB")


def eager_chain():
    chain("A", xa)


timed(eager_chain, "one chain, eager launches   ", "A")

# ---- the price of a graph launch and of a cross-stream hand-off, device side --------------------------------------------------
for k in (1, 5, 20, 100):
    per = n // k
    parts = []
    for j in range(k):
        def part(j=j):
            for i in range(j * per, (j + 1) * per):
                ops.stamp(f"A{i}")
                xa.mul_(1.0001)
            if j == k - 1:
                ops.stamp("Aend")
        parts.append(capture(part, s1))

    def same_stream():
        with torch.cuda.stream(s1):
            for p in parts:
                p.replay()
    timed(same_stream, f"chain as {k:3d} graphs, one stream ", "A")

    evs = [torch.cuda.Event() for _ in range(k)]

    def ping_pong():
        main = torch.cuda.current_stream()
        s1.wait_stream(main)
        s2.wait_stream(main)
        for j, p in enumerate(parts):
            s = s1 if j % 2 == 0 else s2
            if j:
                s.wait_event(evs[j - 1])
            with torch.cuda.stream(s):
                p.replay()
                evs[j].record(s)
        main.wait_event(evs[-1])
    timed(ping_pong, f"chain as {k:3d} graphs, two streams", "A")
