#!/bin/bash
# kernel trace of a few replays of the headline step + the timeline of one replay (tools/trace_timeline.py)
# Usage: bash tools/timeline.sh <tag> [bench args]
set -u
TAG=${1:-tl}; shift || true
OUT=$PWD/gpurun_out/tl_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg "$@" > "$OUT/bench.log" 2>&1
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py "$F" 3 "$OUT/replay.txt" > "$OUT/timeline.txt" 2>&1
cat "$OUT/timeline.txt"; tail -1 "$OUT/bench.log" | cut -c1-200
rm -rf "$OUT/trace"
