#!/bin/bash
# config 4 (OSCC + GraphONE): the precise feature pass forked before / after the training pass's forward chain is created
mkdir -p gpurun_out
B="python bench.py --workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline"
pick() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["ms_per_step"],4))
PY
}
for rep in 1 2; do
EGK_DISABLE=precise_late_fork timeout 300 $B > gpurun_out/v0.json 2>/dev/null; pick gpurun_out/v0.json precise-created-first
timeout 300 $B > gpurun_out/v1.json 2>/dev/null; pick gpurun_out/v1.json precise-created-late
done
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "config4 or c4" > gpurun_out/t_c4.log 2>&1; tail -n 3 gpurun_out/t_c4.log
bash tools/timeline.sh c4b --workload egopack_oscc > /dev/null 2>&1; head -n 7 gpurun_out/tl_c4b/timeline.txt
