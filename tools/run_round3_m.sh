#!/bin/bash
# A/B of two library builds on one box: tools/exp/ab/lib_old.so (EGK_LIB_PATH) vs the in-tree build
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/t_kernels.log 2>&1; tail -n 3 gpurun_out/t_kernels.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline"
pick() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["ms_per_step"],4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2 3; do
EGK_LIB_PATH=$PWD/tools/exp/ab/lib_old.so timeout 300 $B > gpurun_out/ab_old.json 2> gpurun_out/ab_old.err; pick gpurun_out/ab_old.json
timeout 300 $B > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err; pick gpurun_out/ab_new.json
done
EGK_LIB_PATH=$PWD/tools/exp/ab/lib_old.so timeout 300 $B --trn-hidden 4096 > gpurun_out/ab_old4096.json 2>/dev/null; pick gpurun_out/ab_old4096.json
timeout 300 $B --trn-hidden 4096 > gpurun_out/ab_new4096.json 2>/dev/null; pick gpurun_out/ab_new4096.json
EGK_LIB_PATH=$PWD/tools/exp/ab/lib_old.so timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_old.log 2>&1; tail -n 14 gpurun_out/gemm_old.log
timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_new.log 2>&1; tail -n 14 gpurun_out/gemm_new.log
