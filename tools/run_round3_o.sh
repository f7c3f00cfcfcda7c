#!/bin/bash
mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline"
pick() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"],4))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
timeout 300 $B > gpurun_out/p0.json 2>/dev/null; pick gpurun_out/p0.json default
EGK_CHAIN_PRIORITY=-1 timeout 300 $B > gpurun_out/p1.json 2>/dev/null; pick gpurun_out/p1.json chain-high
EGK_WGRAD_PRIORITY=1 timeout 300 $B > gpurun_out/p2.json 2>gpurun_out/p2.err; pick gpurun_out/p2.json wgrad-low
EGK_CHAIN_PRIORITY=-1 EGK_WGRAD_PRIORITY=1 timeout 300 $B > gpurun_out/p3.json 2>/dev/null; pick gpurun_out/p3.json both
done
tail -n 3 gpurun_out/p2.err
