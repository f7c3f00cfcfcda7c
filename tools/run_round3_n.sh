#!/bin/bash
# full GPU suite + the round's evidence set
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu_full.log 2>&1; tail -n 6 gpurun_out/t_gpu_full.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err; tail -c 400 gpurun_out/bench_headline.json
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile.sh r03b > gpurun_out/profile_r03b.log 2>&1; tail -n 8 gpurun_out/profile_r03b.log
