#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
for r in 1 2 3; do
  echo -n "loaders off c3: "; run --gemm-knob 870
  echo -n "loaders on  c3: "; run --gemm-knob 871
done
for r in 1 2; do
  echo -n "loaders off hp4096: "; run --gemm-knob 870 --trn-hidden 4096
  echo -n "loaders on  hp4096: "; run --gemm-knob 871 --trn-hidden 4096
  echo -n "loaders off dry8: "; run --gemm-knob 870 --exchange-dry-run 8
  echo -n "loaders on  dry8: "; run --gemm-knob 871 --exchange-dry-run 8
done
