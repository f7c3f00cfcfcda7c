#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_entrypoints.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/t_part.log 2>&1; grep -n "passed\|failed" gpurun_out/t_part.log | tail -n 3; grep -n "Error\|assert " gpurun_out/t_part.log | head -n 10
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline"
pick() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["ms_per_step"],4))
PY
}
for rep in 1 2 3; do
EGK_DISABLE=late_objective timeout 300 $B > gpurun_out/h0.json 2>/dev/null; pick gpurun_out/h0.json objective-on-chain
timeout 300 $B > gpurun_out/h1.json 2>/dev/null; pick gpurun_out/h1.json objective-late
done
