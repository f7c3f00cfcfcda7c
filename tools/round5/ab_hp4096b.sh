#!/bin/bash
# Hp = 4096 step: EGK_DISABLE toggles, alternating on one box: bash tools/round5/ab_hp4096b.sh "<env A>" "<env B>" [rounds]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; R=${3:-2}
line() { env $1 python bench.py --trn-hidden 4096 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in $(seq $R); do line "$A"; line "$B"; done
