#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2; do
  for k in 1 2 3 4; do echo -n "kchunks $k: "; EGK_WGRAD_KCHUNKS=$k run; done
  for p in 0 24 40; do echo -n "tt pad $p KiB: "; run --gemm-knob $((600+p)); done
done
