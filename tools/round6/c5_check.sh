#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 30 --warmup 8 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
hostname; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for r in 1 2 3; do
  echo -n "c5 default: "; run --workload mtl4 --T 256 --batch 16
  echo -n "c5 loaders off (870): "; run --workload mtl4 --T 256 --batch 16 --gemm-knob 870
  echo -n "c3 default: "; run
  echo -n "c3 loaders off (870): "; run --gemm-knob 870
done
