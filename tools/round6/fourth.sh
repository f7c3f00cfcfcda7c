#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
line() { local tag="$1"; shift; local e="$1"; shift
  env $e python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],4), d['config']['capture'])" || tail -5 $O/$tag.err; }
for i in 1 2; do
line c3_prev_$i "EGK_LIB_PATH=$PWD/tools/exp/build/libegopack_prev.so"
line c3_new_$i "EGK_X=0"
line c3_new_gemmoff_$i "EGK_X=0" --gemm-knob 950
done
line c2_prev "EGK_LIB_PATH=$PWD/tools/exp/build/libegopack_prev.so" --workload ar
line c2_new "EGK_X=0" --workload ar
line c4_prev "EGK_LIB_PATH=$PWD/tools/exp/build/libegopack_prev.so" --workload egopack_oscc
line c4_new "EGK_X=0" --workload egopack_oscc
line c5_prev "EGK_LIB_PATH=$PWD/tools/exp/build/libegopack_prev.so" --workload mtl4 --T 256 --batch 16
line c5_new "EGK_X=0" --workload mtl4 --T 256 --batch 16
