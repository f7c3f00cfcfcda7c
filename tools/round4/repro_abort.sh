#!/bin/bash
# Reproduce the round-3 driver abort: the driver's collection order up to the (then in-process) dist tests, repeated, with
# all-thread faulthandler dumps and C++ stack traces.  Round-3 layout kept as tests/repro_gpu_dist_inproc.py.  The product no
# longer swallows the failed capture, so the exception that preceded the abort is printed.  Writes gpurun_out/repro/*.log.
mkdir -p gpurun_out/repro
rm -f gpurun_out/repro/*
export EGK_ABORT_BT=$PWD/tools/round4/abort_bt.so TORCH_SHOW_CPP_STACKTRACES=1 NCCL_DEBUG=WARN PYTHONFAULTHANDLER=1 EGK_TEST_KEEP_ORDER=1
N=${1:-6}
for i in $(seq 1 $N); do
  s=$(date +%s)
  python3 -X faulthandler -m pytest tests/test_gpu_blockwise.py tests/test_gpu_configs.py tests/repro_gpu_dist_inproc.py -x -q -m gpu -p no:cacheprovider \
      > gpurun_out/repro/run_$i.log 2>&1
  rc=$?
  echo "run $i rc $rc $(( $(date +%s) - s )) s $(tail -1 gpurun_out/repro/run_$i.log | cut -c1-200)" | tee -a gpurun_out/repro/summary.txt
done
