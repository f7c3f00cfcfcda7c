#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > $O/gemm_tests.log 2>&1; tail -5 $O/gemm_tests.log
python tools/round6/r192_bench.py > $O/r192_bench.txt 2>&1; cat $O/r192_bench.txt
line() { local tag="$1"; shift; local e="$1"; shift
  env $e python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>$O/$tag.err | tail -1 > $O/$tag.json
  python -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],4), d['config']['capture'])" || tail -5 $O/$tag.err; }
for i in 1 2; do
line c3_r192_$i "EGK_X=0"
line c3_r96_$i "EGK_X=0" --gemm-knob 900
done
line c5_r192 "EGK_X=0" --workload mtl4 --T 256 --batch 16
line c5_r96 "EGK_X=0" --workload mtl4 --T 256 --batch 16 --gemm-knob 900
