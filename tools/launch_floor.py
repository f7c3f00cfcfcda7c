#!/usr/bin/env python3
"""Per-node cost of dependent tiny kernels replayed from a hipGraph (the launch floor every small kernel pays)."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

x = torch.zeros(256, device="cuda")
y = torch.zeros(256, device="cuda")


from egopack_amd import _lib
lib = _lib.load()


def chain(n):
    for _ in range(n):
        lib.egk_axpby(ops._stream(), ops._p(x), ops._p(y), ops._p(y), 256, 1.0, 1.0)


def time_graph(fn, n):
    fn(2)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(2)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn(n)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("egk tiny kernel  : %.2f us per node" % time_graph(chain, 400))
print("torch add_ tiny  : %.2f us per node" % time_graph(lambda n: [y.add_(x) for _ in range(n)], 400))
