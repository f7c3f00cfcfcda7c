#!/bin/bash
# every default-on switch off, one at a time, against the default on the same box:  bash tools/round6/switch_sweep.sh [bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 4))" 2>/dev/null || echo failed; }
names=$(python -c "
from egopack_amd import switches
print(' '.join(k for k, (d, _) in switches.REGISTRY.items() if d))")
echo "default: $(run "$@")"
i=0
for n in $names; do
  echo "$n: $(EGK_DISABLE=$n run "$@")"
  i=$((i+1)); if [ $((i % 10)) -eq 0 ]; then echo "default: $(run "$@")"; fi
done
echo "default: $(run "$@")"
