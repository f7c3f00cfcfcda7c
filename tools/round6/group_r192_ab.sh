#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
for r in 1; do
  echo -n "group r192 on  c3: "; run
  echo -n "group r192 off c3: "; run --gemm-knob 860
done
for r in 1 2; do
  for w in "--workload egopack_oscc" "--workload ar"; do
    echo -n "on  [$w]: "; run $w
    echo -n "off [$w]: "; run $w --gemm-knob 860
  done
done
