#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_step_structures.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', round(d['ms_per_step'],4))"
