#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_full2; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
line() { local tag="$1"; shift; local e="$1"; shift
  env $e python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],4), d['config']['capture'], d['config'].get('exchange_graph'))" || tail -5 $O/$tag.err; }
line c3 "EGK_X=0"
line dry8_auto "EGK_X=0" --exchange-dry-run 8
line dry8_staged "EGK_X=0" --exchange-dry-run 8 --exchange-graph staged
line dry8_sharded_one "EGK_ENABLE=sharded_update" --exchange-dry-run 8 --exchange-graph one
line dry8_sharded_staged "EGK_ENABLE=sharded_update" --exchange-dry-run 8 --exchange-graph staged
line dry8_sharded_auto "EGK_ENABLE=sharded_update" --exchange-dry-run 8
