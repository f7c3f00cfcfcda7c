#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2 3; do
  echo -n "tall on  c3: "; run
  echo -n "tall off c3: "; run --gemm-knob 850
done
for r in 1 2; do
  echo -n "tall on  dry8: "; run --exchange-dry-run 8
  echo -n "tall off dry8: "; run --exchange-dry-run 8 --gemm-knob 850
  echo -n "tall on  c5: "; run --workload mtl4 --T 256 --batch 16
  echo -n "tall off c5: "; run --workload mtl4 --T 256 --batch 16 --gemm-knob 850
  echo -n "tall on  c2: "; run --workload ar
  echo -n "tall off c2: "; run --workload ar --gemm-knob 850
done
