import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_configs as T
from egopack_amd import ops
def run(captured, env=None):
    if env: os.environ["EGK_DISABLE"] = env
    else: os.environ.pop("EGK_DISABLE", None)
    torch.manual_seed(0)
    args, step, opt, dev, merged, modules, sds, weights = T._build("c4_egopack_oscc_K4096_d3", "bf16")
    if captured:
        step.capture(dev, merged, warmup=2)
        for _ in range(2): step.replay()
    else:
        for _ in range(4): step.step(dev, merged)
    torch.cuda.synchronize()
    return opt.flat_p.clone()
for env in (None, "one_pass"):
    a, b = run(False, env), run(True, env)
    d = (a - b).abs()
    print(env, "eager vs captured: max", float(d.max()), "n differing", int((d > 0).sum()), "of", d.numel(), flush=True)
    a2 = run(False, env)
    print(env, "eager vs eager: n differing", int(((a - a2).abs() > 0).sum()), flush=True)
