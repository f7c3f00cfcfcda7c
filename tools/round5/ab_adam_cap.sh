#!/bin/bash
# Adam workgroup cap (egk_tune 6) in configs 4 and 3, alternating on one box
cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg --egk-tune 6=$2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 cap=$2', d['ms_per_step'])"; }
for r in 1 2; do for cap in 4096 2048 1024 512 256; do run "--workload egopack_oscc" $cap; done; done
for r in 1 2; do for cap in 4096 2048 1024 512; do run "" $cap; done; done
