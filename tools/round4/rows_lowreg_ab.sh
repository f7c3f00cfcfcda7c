#!/bin/bash
# the backward pass's row kernels under 96 registers (co-resident with two weight-gradient workgroups per CU): egk_tune 3 = 0 / 1
python3 -m pytest tests/test_gpu_kernels.py -q -x -k "rows1024 or segment_statistics" 2>&1 | tail -3
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2 3; do
  run c3_v2_$rep python3 bench.py $C
  run c3_generic_$rep python3 bench.py $C --egk-tune 3=0
done
run c2_v2 python3 bench.py $C --workload ar
run c2_generic python3 bench.py $C --workload ar --egk-tune 3=0
run c4_v2 python3 bench.py $C --workload egopack_oscc
run c4_generic python3 bench.py $C --workload egopack_oscc --egk-tune 3=0
run c5_v2 python3 bench.py $C --workload mtl4 --T 256 --batch 16
run c5_generic python3 bench.py $C --workload mtl4 --T 256 --batch 16 --egk-tune 3=0
bash tools/timeline.sh lowreg > /dev/null 2>&1; grep -E "csr_gather|graphln_bwd|rowln_bwd" gpurun_out/tl_lowreg/replay.txt | cut -c1-70
