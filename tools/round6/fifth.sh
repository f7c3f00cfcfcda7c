#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
line() { local tag="$1"; shift; local e="$1"; shift
  env $e python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],4), d['config']['capture'])" || tail -5 $O/$tag.err; }
for i in 1 2; do
line c3_base_$i "EGK_X=0"
line c3_w1536_$i "EGK_X=0" --egk-tune 2=1536
line c3_w1536_p1024_$i "EGK_X=0" --egk-tune 2=1536,1=1024
line c3_w2048_p1536_$i "EGK_X=0" --egk-tune 2=2048,1=1536
line c3_w384_$i "EGK_X=0" --egk-tune 2=384,1=384
done
