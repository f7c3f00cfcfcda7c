#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2; do
for c in 6 8; do
  echo -n "count $c c2: "; EGK_WGRAD_COUNT=$c run --workload ar
  echo -n "count $c c5: "; EGK_WGRAD_COUNT=$c run --workload mtl4
  echo -n "count $c hp4096: "; EGK_WGRAD_COUNT=$c run --trn-hidden 4096
  echo -n "count $c dry8: "; EGK_WGRAD_COUNT=$c run --exchange-dry-run 8
  echo -n "count $c dry8 staged: "; EGK_WGRAD_COUNT=$c run --exchange-dry-run 8 --exchange-graph staged
done; done
