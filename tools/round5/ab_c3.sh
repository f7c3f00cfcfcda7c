#!/bin/bash
# same-box alternating A/B of the headline (config 3): bash tools/round5/ab_c3.sh "<env A>" "<env B>" [rounds]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; R=${3:-2}
line() { local e="${1%% -- *}"; local f=""; [[ "$1" == *" -- "* ]] && f="${1#* -- }"
  env $e python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in $(seq $R); do line "$A"; line "$B"; done
