#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bit_equal_to_generic or segment_statistics" > $O/gemm_tests.log 2>&1; tail -3 $O/gemm_tests.log
for v in 8 16 17 18; do python tools/gemm_stamps.py --phases 6144 1024 1024 0 0 1 $v 2>&1 | grep -v amdgpu.ids; done > $O/phases_k1024.txt; cat $O/phases_k1024.txt
for v in 8 16 18; do python tools/gemm_stamps.py --phases 6144 1024 4608 0 0 1 $v 2>&1 | grep -v amdgpu.ids; done > $O/phases_k4608.txt; cat $O/phases_k4608.txt
python tools/round6/r192_bench.py --variants 8,16,17,18 --rows 6144 > $O/r192_bench.txt 2>&1; cat $O/r192_bench.txt
line() { local tag="$1"; shift; local e="$1"; shift
  env $e python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],4), d['config']['capture'])" || tail -5 $O/$tag.err; }
for i in 1 2; do
line c3_k900_$i "EGK_X=0" --gemm-knob 900
line c3_k901_$i "EGK_X=0" --gemm-knob 901
line c3_k902_$i "EGK_X=0" --gemm-knob 902
line c3_k903_$i "EGK_X=0" --gemm-knob 903
done
