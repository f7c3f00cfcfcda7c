#!/bin/bash
# the round's bench lines (driver protocol) for every BASELINE configuration that fits one GPU + the full GPU suite
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/t_gpu_full_r4c.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/t_gpu_full_r4c.log
python3 bench.py > gpurun_out/bench_r04_c3.json 2> gpurun_out/bench_r04_c3.err
python3 bench.py --workload ar > gpurun_out/bench_r04_c2.json 2> gpurun_out/bench_r04_c2.err
python3 bench.py --workload egopack_oscc > gpurun_out/bench_r04_c4.json 2> gpurun_out/bench_r04_c4.err
python3 bench.py --workload mtl4 --T 256 --batch 16 > gpurun_out/bench_r04_c5.json 2> gpurun_out/bench_r04_c5.err
python3 bench.py --exchange-dry-run 8 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r04_dry8_auto.json 2> gpurun_out/bench_r04_dry8_auto.err
python3 bench.py --exchange-dry-run 8 --exchange-graph staged --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r04_dry8_staged.json 2> gpurun_out/bench_r04_dry8_staged.err
for f in c2 c3 c4 c5 dry8_auto dry8_staged; do python3 - gpurun_out/bench_r04_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rl=d.get('roofline') or {}
    print(sys.argv[1], round(d['ms_per_step'],4), round(d['value']), rl.get('kernel'), round(rl.get('frac',0),3), 'traffic', rl.get('traffic'), d['config'].get('capture'), d['config'].get('exchange_graph'))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
