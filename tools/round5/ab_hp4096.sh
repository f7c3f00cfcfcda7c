#!/bin/bash
# the workgroup-per-row LayerNorm kernels (egk_tune 7) on the Hp = 4096 step, alternating on one box
cd $GRAFT_REPO_ROOT
run() { python bench.py --trn-hidden 4096 $1 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg --egk-tune 7=$2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 wide=$2', d['ms_per_step'])"; }
for r in 1 2; do run "" 1; run "" 0; done
for r in 1; do run "--workload ar" 1; run "--workload ar" 0; done
