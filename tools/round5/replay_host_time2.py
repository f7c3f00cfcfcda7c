#!/usr/bin/env python3
"""Does something queued on the launch stream in front of a replay make the graph launch wait on the host?  Host time of replay()
with nothing / a torch device-to-device copy_ / a library kernel launch / an H2D copy from pinned memory in between."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch

import test_gpu_configs as T
from egopack_amd import ops

NAME = sys.argv[1] if len(sys.argv) > 1 else "c3_mtl_B64_T32"
args, step, opt, dev, merged, modules, sds, weights = T._build(NAME, "bf16")
step.capture(dev, merged, warmup=2)
x = merged.x if merged is not None else next(iter(dev.values())).x
src = x.clone()
pinned = torch.empty(x.shape, dtype=x.dtype).pin_memory()
side = torch.cuda.Stream()
for mode in ("nothing", "torch copy_ d2d", "library cast kernel", "pinned H2D on main", "pinned H2D on side + wait_stream", "d2d on side + wait_stream"):
    for _ in range(3):
        step.replay()
    torch.cuda.synchronize()
    host, n = [], 40
    t0 = time.perf_counter()
    for _ in range(n):
        if mode == "torch copy_ d2d":
            x.copy_(src, non_blocking=True)
        elif mode == "library cast kernel":
            ops.cast_raw(src, torch.float32)
        elif mode == "pinned H2D on main":
            x.copy_(pinned, non_blocking=True)
        elif mode == "pinned H2D on side + wait_stream":
            with torch.cuda.stream(side):
                src.copy_(pinned, non_blocking=True)
            torch.cuda.current_stream().wait_stream(side)
        elif mode == "d2d on side + wait_stream":
            with torch.cuda.stream(side):
                src.copy_(x, non_blocking=True)
            torch.cuda.current_stream().wait_stream(side)
        a = time.perf_counter()
        step.replay()
        host.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n * 1e3
    host.sort()
    print(f"{mode:36s}: {total:.3f} ms/step, host time of replay() median {host[n // 2] * 1e3:.3f} ms", flush=True)
