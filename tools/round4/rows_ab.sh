#!/bin/bash
mkdir -p gpurun_out/rows
for rep in 1 2; do
  EGK_ROWS_V2=0 python3 tools/row_bench.py bf16 > gpurun_out/rows/generic_$rep.txt 2>&1
  EGK_ROWS_V2=1 python3 tools/row_bench.py bf16 > gpurun_out/rows/v2_$rep.txt 2>&1
done
paste gpurun_out/rows/generic_1.txt gpurun_out/rows/v2_1.txt | cut -c1-200
echo; paste gpurun_out/rows/generic_2.txt gpurun_out/rows/v2_2.txt | cut -c1-200
python3 -m pytest tests/test_gpu_kernels.py -q -x -k "gather_in_the_contraction or rows1024" -p no:cacheprovider 2>&1 | tail -15
for rep in 1 2; do
for f in on off; do
  if [ $f = off ]; then export EGK_DISABLE=gather_fusion; else unset EGK_DISABLE; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gather_fusion $f', round(d['ms_per_step'],4), d['block_ms_min'])"
done; done
unset EGK_DISABLE
bash tools/timeline.sh r4a > gpurun_out/rows/timeline.txt 2>&1; head -80 gpurun_out/tl_r4a/replay.txt
