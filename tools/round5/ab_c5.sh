#!/bin/bash
# same-box alternating A/B of config 5 (mtl4, T = 256, batch 16): bash tools/round5/ab_c5.sh "<env A>" "<env B>" [rounds]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; R=${3:-2}
line() { env $1 python bench.py --workload mtl4 --T 256 --batch 16 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in $(seq $R); do line "$A"; line "$B"; done
