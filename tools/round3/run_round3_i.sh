#!/bin/bash
# N-rank dry run (8 ranks' exchange path on one GPU, 1-rank RCCL group): one graph incl. the exchange vs the staged graphs
mkdir -p gpurun_out
B="python bench.py --exchange-dry-run 8 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline"
pick() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"],4), d["config"].get("capture"), d["config"].get("capture_fallbacks"), d.get("exchange"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
timeout 300 $B > gpurun_out/dry8_onegraph.json 2> gpurun_out/dry8_onegraph.err; pick gpurun_out/dry8_onegraph.json || tail -n 20 gpurun_out/dry8_onegraph.err
EGK_DISABLE=exchange_early_adam timeout 300 $B > gpurun_out/dry8_onegraph_noearly.json 2> gpurun_out/dry8_onegraph_noearly.err; pick gpurun_out/dry8_onegraph_noearly.json
timeout 300 $B --grad-compress bf16 > gpurun_out/dry8_onegraph_bf16.json 2> gpurun_out/dry8_onegraph_bf16.err; pick gpurun_out/dry8_onegraph_bf16.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline > gpurun_out/one_rank.json 2>/dev/null; pick gpurun_out/one_rank.json
done
tail -n 5 gpurun_out/dry8_onegraph.err
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_two_rank.py -x -q -m gpu > gpurun_out/t_dist.log 2>&1; grep -n "passed\|failed" gpurun_out/t_dist.log
