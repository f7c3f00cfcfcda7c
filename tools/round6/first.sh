#!/bin/bash
# round 6, first GPU call: new tests, the N-rank structure tests, and dry-run-8 lines with / without the stored gradient slots
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
O=gpurun_out/r6a
python -m pytest tests/test_gpu_meters.py -x -q -m gpu > $O/meters.log 2>&1; tail -3 $O/meters.log
python -m pytest tests/test_gpu_dist.py tests/test_gpu_two_rank.py tests/test_gpu_step_structures.py -x -q -m gpu > $O/dist.log 2>&1; tail -5 $O/dist.log
line() { local tag="$1"; shift; local e="$1"; shift
  env $e python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],4), d['config']['capture'], d['config'].get('exchange_graph'))" || tail -5 $O/$tag.err; }
for i in 1 2; do
line c3_$i "EGK_X=0"
line dry8_staged_$i "EGK_X=0" --exchange-dry-run 8 --exchange-graph staged
line dry8_staged_nostore_$i "EGK_DISABLE=grad_store" --exchange-dry-run 8 --exchange-graph staged
line dry8_one_$i "EGK_X=0" --exchange-dry-run 8 --exchange-graph one
line dry8_one_nostore_$i "EGK_DISABLE=grad_store" --exchange-dry-run 8 --exchange-graph one
done
line dry8_auto "EGK_X=0" --exchange-dry-run 8
line dry8_sharded "EGK_ENABLE=sharded_update" --exchange-dry-run 8
