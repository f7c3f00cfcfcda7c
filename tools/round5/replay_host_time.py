#!/usr/bin/env python3
"""Host time of one replay of the captured step (the call returns when the runtime has enqueued the graph) against the device
time per step: a training loop is host-bound when its own per-step host work + this exceeds the device time."""
import argparse
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch

import test_gpu_configs as T

for name in ("c3_mtl_B64_T32", "c4_egopack_oscc_K4096_d3"):
    args, step, opt, dev, merged, modules, sds, weights = T._build(name, "bf16")
    step.capture(dev, merged, warmup=2)
    for _ in range(5):
        step.replay()
    torch.cuda.synchronize()
    n = 50
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        step.replay()
        host.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n * 1e3
    host.sort()
    # one replay at a time: launch, then wait
    lone = []
    for _ in range(10):
        a = time.perf_counter()
        step.replay()
        b = time.perf_counter()
        torch.cuda.synchronize()
        lone.append((b - a) * 1e3)
    print(f"{name}: back-to-back {total:.3f} ms/step; host time of replay() median {host[n // 2] * 1e3:.3f} ms (min {host[0] * 1e3:.3f}); "
          f"a lone replay() returns after {sorted(lone)[5]:.3f} ms", flush=True)
