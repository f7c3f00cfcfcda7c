#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
for r in 1 2; do
  for w in "" "--workload ar" "--workload egopack_oscc" "--workload mtl4 --T 256 --batch 16" "--trn-hidden 4096"; do
    echo -n "off [$w]: "; run $w
    echo -n "on  [$w]: "; run $w --gemm-knob 881
  done
done
