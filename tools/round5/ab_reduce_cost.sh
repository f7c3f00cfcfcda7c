#!/bin/bash
# the split-K policy's fixed cost of the reduce launch (egk_gemm_set_pipeline(400 + 10 x us); default 3.5 us) in configs 2, 4, 3
cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg --gemm-knob $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 knob=$2', d['ms_per_step'])"; }
for r in 1 2; do for k in 435 460 480 520; do run "--workload ar" $k; done; done
for r in 1 2; do for k in 435 460 480 520; do run "--workload egopack_oscc" $k; done; done
for r in 1; do for k in 435 480; do run "" $k; done; done
