#!/usr/bin/env python3
"""Host time of one graph replay (hipGraphLaunch returns when every node has been handed to its queue) against the device
time of the step: is a configuration bound by the rate at which a replay's nodes reach the hardware queues?

    python tools/round4/replay_host_time.py [bench.py workload arguments]
"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    args = bench.parse_args(sys.argv[1:] + ["--no-cpu-baseline", "--no-roofline", "--no-f32-leg"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hold = {}
    orig = bench.time.perf_counter
    # reuse bench.measure up to the capture by running it with 1 step, then grab the step object through a hook
    from egopack_amd import engine
    cap = engine.StepBase.capture

    def spy(self, *a, **k):
        hold["step"] = self
        return cap(self, *a, **k)
    engine.StepBase.capture = spy
    res = bench.measure(args, 0, 1, dev, 5, 3, want_roofline=False, want_cpu=False)
    step = hold["step"]
    for _ in range(5):
        step.replay()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step.replay()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    # one replay alone: the host call, then the wait
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    g = getattr(step, "_graph", None)
    if hasattr(g, "segments"):
        segs = g.segments()
        print("plan:", g.info())
        print("segments (nodes, stream, waits, records):", " ".join(f"{a}/{b}/{c}/{d}" for a, b, c, d in segs))
    print(f"workload {args.workload}: bench {res['ms']:.3f} ms/step; {n} replays back to back: host {t_host * 1e3:.3f} ms per call, "
          f"{t_all * 1e3:.3f} ms per step incl. the final wait; one replay alone: host call {1e3 * (t1 - t0):.3f} ms, done after {1e3 * (t2 - t0):.3f} ms")


if __name__ == "__main__":
    main()
