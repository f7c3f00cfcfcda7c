#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "f32" > gpurun_out/t_f32.log 2>&1; tail -n 12 gpurun_out/t_f32.log
B="python bench.py --compute f32 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
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
EGK_DISABLE=f32_wgrad_groups timeout 300 $B > gpurun_out/f32_off.json 2> gpurun_out/f32_off.err; pick gpurun_out/f32_off.json singles
timeout 300 $B > gpurun_out/f32_on.json 2> gpurun_out/f32_on.err; pick gpurun_out/f32_on.json grouped
done
tail -n 3 gpurun_out/f32_on.err
