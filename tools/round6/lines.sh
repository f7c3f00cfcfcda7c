#!/bin/bash
# bench lines of every configuration on one box (no roofline / CPU legs)
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('capture'))"; }
echo -n "c3: "; run
echo -n "c2: "; run --workload ar
echo -n "c4: "; run --workload egopack_oscc
echo -n "c5: "; run --workload mtl4 --T 256 --batch 16
echo -n "hp4096: "; run --trn-hidden 4096
echo -n "dry8 auto: "; run --exchange-dry-run 8
echo -n "dry8 staged: "; run --exchange-dry-run 8 --exchange-graph staged
echo -n "dry8 sharded: "; EGK_ENABLE=sharded_update run --exchange-dry-run 8
