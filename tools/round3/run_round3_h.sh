#!/bin/bash
# Hp = 4096 bench lines, the 2-rank one-GPU bench path with the new timed blocks / exchange fields, N-rank dry run
mkdir -p gpurun_out
python bench.py --trn-hidden 4096 --steps 20 --warmup 5 > gpurun_out/bench_hp4096.json 2> gpurun_out/bench_hp4096.err; tail -c 2500 gpurun_out/bench_hp4096.json
python bench.py --trn-hidden 4096 --workload ar --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_hp4096_ar.json 2> gpurun_out/bench_hp4096_ar.err; tail -c 1200 gpurun_out/bench_hp4096_ar.json
python bench.py --gpus 2 --one-gpu-gloo --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err; tail -c 1500 gpurun_out/bench_2rank_gloo.json; tail -n 5 gpurun_out/bench_2rank_gloo.err
python bench.py --exchange-dry-run 8 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_dry8.json 2> gpurun_out/bench_dry8.err; tail -c 1500 gpurun_out/bench_dry8.json
