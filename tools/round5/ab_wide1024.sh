#!/bin/bash
# the workgroup-per-row LayerNorm kernels also for rows of 1024 columns (egk_tune 7 = 2) against the one-wave kernels (1): c3, c5
cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg --egk-tune 7=$2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 tune7=$2', d['ms_per_step'])"; }
for r in 1 2 3; do run "" 1; run "" 2; done
for r in 1 2; do run "--workload mtl4 --T 256 --batch 16" 1; run "--workload mtl4 --T 256 --batch 16" 2; done
