#!/bin/bash
mkdir -p gpurun_out
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
for c in 2 3 4 6 8; do
EGK_F32_WGRAD_COUNT=$c timeout 300 $B > gpurun_out/f32_c$c.json 2> gpurun_out/f32_c$c.err; pick gpurun_out/f32_c$c.json count$c
done
EGK_DISABLE=f32_wgrad_groups timeout 300 $B > gpurun_out/f32_off.json 2> gpurun_out/f32_off.err; pick gpurun_out/f32_off.json singles
done
