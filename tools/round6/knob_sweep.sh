#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
for r in 1 2; do
  echo -n "default: "; run
  for k in 200 300 900 950 850 101 102 104 108 501; do echo -n "gemm-knob $k: "; run --gemm-knob $k; done
  echo -n "live share 0.5: "; EGK_LIVE_SHARE=0.5 run
  echo -n "rows v2 off: "; EGK_ROWS_V2=0 run
  echo -n "default: "; run
done
