"""cProfile of the STEADY part of an entry point's training loop: the profiler is switched on at the 40th train_step call
(eager steps, capture and the dataset tables are behind) and off when the entry point returns.
  python tools/round6/loop_profile.py main_temporal.py <overrides...>  ->  gpurun_out/r06_<entry>_loop_cprofile.txt"""
import cProfile
import io
import os
import pstats
import runpy
import sys

ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from egopack_amd import engine  # noqa: E402

entry = sys.argv[1]
prof, n = cProfile.Profile(), [0]
owner = next(c for c in engine.MTLStep.__mro__ if "train_step" in c.__dict__)
orig = owner.__dict__["train_step"]


def ts(self, *a, **k):
    n[0] += 1
    if n[0] == 40:
        prof.enable()
    return orig(self, *a, **k)


owner.train_step = ts
sys.argv = [entry, *sys.argv[2:]]
try:
    import atexit
    runpy.run_path(entry, run_name="__main__")
finally:
    prof.disable()
    out = io.StringIO()
    pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(60)
    os.makedirs(f"{ROOT}/gpurun_out", exist_ok=True)
    with open(f"{ROOT}/gpurun_out/r06_{os.path.splitext(os.path.basename(entry))[0]}_loop_cprofile.txt", "w") as f:
        f.write(f"steps profiled: {n[0] - 39}\n" + out.getvalue())
