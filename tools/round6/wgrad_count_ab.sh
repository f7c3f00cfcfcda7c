#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for c in 6 8 4 5 7; do echo -n "count $c: "; EGK_WGRAD_COUNT=$c python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done
