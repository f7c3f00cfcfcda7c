#!/bin/bash
# full GPU suite + the round's evidence set
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu_full.log 2>&1; grep -n "passed\|failed" gpurun_out/t_gpu_full.log | tail -n 2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err; tail -c 300 gpurun_out/bench_headline.json
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile.sh r03c > gpurun_out/profile_r03c.log 2>&1; tail -n 3 gpurun_out/profile_r03c.log
bash tools/timeline.sh r03c > gpurun_out/timeline_r03c.log 2>&1; head -n 6 gpurun_out/tl_r03c/timeline.txt
