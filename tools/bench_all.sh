#!/bin/bash
# All bench workloads, short runs, one line each.  Usage: bash tools/bench_all.sh [extra bench args]
for w in "mtl" "egopack_oscc" "ar" "mtl4"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --workload $w "$@" 2>/dev/null | tail -1 | python tools/ms.py "$w"
done
