#!/usr/bin/env python3
"""When does an independent branch of a replayed hipGraph start?  Two chains with no edge between them (A on the capture stream,
B forked from the capture's first event), every link = one clock stamp (egk_stamp, one lane) + one elementwise launch; the stamps
say when each link ran inside the replay.  Config 4's precise pass is such a branch B.
Usage: python3 tools/round4/branch_start_probe.py [--a 20x2000000] [--b 40x200000] [--swap] [--mid]"""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--a", default="20x4000000", help="chain A: links x elements of the f32 tensor each link scales")
ap.add_argument("--b", default="40x400000")
ap.add_argument("--swap", action="store_true", help="issue B before A (creation order)")
ap.add_argument("--origin-b", action="store_true", help="B runs on the capture stream, A on the forked one")
ap.add_argument("--root", action="store_true", help="a first launch on the capture stream that both chains follow")
args = ap.parse_args()
na, ea = (int(v) for v in args.a.split("x"))
nb, eb = (int(v) for v in args.b.split("x"))
dev = "cuda"
xa, xb, x0 = torch.ones(ea, device=dev), torch.ones(eb, device=dev), torch.ones(1024, device=dev)
ops.stamps_enable(dev, slots=512)
side = torch.cuda.Stream()
cap = torch.cuda.Stream()


def chain(name, n, x):
    for i in range(n):
        ops.stamp(f"{name}{i}")
        x.mul_(1.0001)
    ops.stamp(f"{name}end")


def body():
    main = torch.cuda.current_stream()
    if args.root:
        x0.mul_(1.0001)
    side.wait_stream(main)
    first, second = (("B", nb, xb), ("A", na, xa)) if args.swap else (("A", na, xa), ("B", nb, xb))
    for name, n, x in (first, second):
        on_side = (name == "B") != args.origin_b
        if on_side:
            with torch.cuda.stream(side):
                chain(name, n, x)
        else:
            chain(name, n, x)
    main.wait_stream(side)
    x0.mul_(1.0001)


with torch.cuda.stream(cap):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
    body()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
st = dict(ops.stamps_read())
a0, b0 = st["A0"], st["B0"]
print(f"A: {na} links, first stamp {a0:.1f} us, end {st['Aend']:.1f} us ({(st['Aend'] - a0) / na:.1f} us / link)")
print(f"B: {nb} links, first stamp {b0:.1f} us, end {st['Bend']:.1f} us ({(st['Bend'] - b0) / nb:.1f} us / link)")
print(f"host: hipGraphLaunch returned after {(t1 - t0) * 1e6:.0f} us, replay done after {(t2 - t0) * 1e6:.0f} us")
