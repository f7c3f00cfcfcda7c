#!/bin/bash
# the whole GPU suite + one bench line per BASELINE configuration (round 6)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_full; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
line() { local tag="$1"; shift
  python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],4), d['config']['capture'])" || tail -5 $O/$tag.err; }
line c3
line c2 --workload ar
line c4 --workload egopack_oscc
line c5 --workload mtl4 --T 256 --batch 16
line hp4096 --trn-hidden 4096
line dry8_auto --exchange-dry-run 8
line dry8_staged --exchange-dry-run 8 --exchange-graph staged
