#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2; do
for c in 4 5 6 7 8; do
  echo -n "count $c c2: "; EGK_WGRAD_COUNT=$c run --workload ar
  echo -n "count $c c4: "; EGK_WGRAD_COUNT=$c run --workload egopack_oscc
  echo -n "count $c c5: "; EGK_WGRAD_COUNT=$c run --workload mtl4 --T 256 --batch 16
done; done
