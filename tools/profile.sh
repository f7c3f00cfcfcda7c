#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + two separate PMC passes of bench.py.
# Usage: bash tools/profile.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py $ARGS > "$OUT/bench_stats.log" 2>&1
echo "stats rc=$?"
PARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32-leg --mode eager $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 bench.py $PARGS > "$OUT/bench_fetch.log" 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python3 bench.py $PARGS > "$OUT/bench_write.log" 2>&1
echo "write rc=$?"
# matrix-pipe utilisation: SQ counters in a pass of their own (8 SQ slots per pass; never combined with a trace domain)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_mfma" -o bench -- python3 bench.py $PARGS > "$OUT/bench_mfma.log" 2>&1
echo "mfma rc=$?"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.md" 2>&1
echo "summary rc=$?"
find "$OUT" -name "*.csv" | head -20
# keep the merged payload small: drop the raw per-dispatch traces, keep stats + counters summaries
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
