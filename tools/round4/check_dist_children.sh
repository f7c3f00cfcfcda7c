#!/bin/bash
# The process-group tests in their round-4 layout (child processes), repeated.
mkdir -p gpurun_out/distchild
N=${1:-3}
for i in $(seq 1 $N); do
  python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_step_structures.py -q -m gpu -p no:cacheprovider > gpurun_out/distchild/run_$i.log 2>&1
  echo "distchild $i rc $? $(tail -1 gpurun_out/distchild/run_$i.log | cut -c1-200)" | tee -a gpurun_out/distchild/summary.txt
done
