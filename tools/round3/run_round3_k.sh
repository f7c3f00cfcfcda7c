#!/bin/bash
mkdir -p gpurun_out
B="python bench.py --exchange-dry-run 8 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline"
pick() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"],4), d["config"].get("capture"), d["config"].get("capture_fallbacks"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
timeout 300 $B > gpurun_out/dry8_onegraph.json 2> gpurun_out/dry8_onegraph.err; pick gpurun_out/dry8_onegraph.json
EGK_ENABLE=sharded_update timeout 300 $B > gpurun_out/dry8_sharded.json 2> gpurun_out/dry8_sharded.err; pick gpurun_out/dry8_sharded.json
timeout 300 $B --staged off > gpurun_out/dry8_onepiece.json 2> gpurun_out/dry8_onepiece.err; pick gpurun_out/dry8_onepiece.json
done
tail -n 3 gpurun_out/dry8_sharded.err
timeout 900 python -m pytest tests/test_gpu_two_rank.py -x -q -m gpu > gpurun_out/t_two.log 2>&1; tail -n 15 gpurun_out/t_two.log | cut -c1-600
