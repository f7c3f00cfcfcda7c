#!/bin/bash
# live-loader loop after the one-buffer replay path + the tests that cover train_step / staging
mkdir -p gpurun_out
python -m egopack_amd.build > gpurun_out/build.log 2>&1 || { tail -n 20 gpurun_out/build.log; exit 1; }
python tools/train_loop_bench.py --live --workers=0 > gpurun_out/loop_live0.log 2>&1; tail -n 6 gpurun_out/loop_live0.log
python tools/train_loop_bench.py --live --workers=0 --profile > gpurun_out/loop_live0_prof.log 2>&1; tail -n 45 gpurun_out/loop_live0_prof.log
python -m pytest tests/test_gpu_entrypoints.py tests/test_gpu_feature_store.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/t_loop.log 2>&1; tail -n 8 gpurun_out/t_loop.log
