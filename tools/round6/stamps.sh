#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_stamps; mkdir -p $O
for v in 8 16; do for K in 1024 2048 4608; do python tools/gemm_stamps.py --phases 6144 1024 $K 0 0 1 $v 2>&1 | grep -v "amdgpu.ids"; done; done | tee $O/phases_nn.txt
for v in 8 16; do python tools/gemm_stamps.py --phases 6144 1024 1024 0 1 1 $v 2>&1 | grep -v "amdgpu.ids"; done | tee $O/phases_nt.txt
python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_kernels.py --deselect tests/test_gpu_models.py > $O/gpu_tests_rest.log 2>&1; tail -4 $O/gpu_tests_rest.log
