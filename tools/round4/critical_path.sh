#!/bin/bash
# Usage: bash tools/round4/critical_path.sh <tag> [bench args]
TAG=$1; shift
OUT=$PWD/gpurun_out/cp_$TAG
mkdir -p "$OUT"; cd "$OUT"; rm -f graph_*_dot_print_*
export TMPDIR=/tmp DEBUG_HIP_GRAPH_DOT_PRINT=1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o bench -- python3 ../../bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg "$@" > "$OUT/bench.log" 2>&1
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
D=$(ls -S graph_*_dot_print_* | head -1)
cd ../..
python3 tools/graph_critical_path.py "$OUT/$D" "$F" 3 > "$OUT/critical_path.txt" 2>&1
cat "$OUT/critical_path.txt"
cp "$F" "$OUT/kernel_trace.csv"; rm -rf "$OUT/trace"
