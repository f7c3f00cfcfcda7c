#!/bin/bash
# the round's bench lines (driver protocol) for every BASELINE configuration that fits one GPU
set -u
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_r05_c3.json 2> gpurun_out/bench_r05_c3.err
python3 bench.py --workload ar > gpurun_out/bench_r05_c2.json 2> gpurun_out/bench_r05_c2.err
python3 bench.py --workload egopack_oscc > gpurun_out/bench_r05_c4.json 2> gpurun_out/bench_r05_c4.err
python3 bench.py --workload mtl4 --T 256 --batch 16 > gpurun_out/bench_r05_c5.json 2> gpurun_out/bench_r05_c5.err
python3 bench.py --trn-hidden 4096 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r05_hp4096_mtl.json 2> gpurun_out/bench_r05_hp4096_mtl.err
python3 bench.py --exchange-dry-run 8 --exchange-graph staged --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r05_dry8_staged.json 2> gpurun_out/bench_r05_dry8_staged.err
for f in c2 c3 c4 c5 hp4096_mtl dry8_staged; do python3 - gpurun_out/bench_r05_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rl=d.get('roofline') or {}; cb=d.get('cpu_baseline') or {}
    print(sys.argv[1], round(d['ms_per_step'],4), round(d['value']), rl.get('kernel'), round(rl.get('frac',0),3), 'step', round((rl.get('step') or {}).get('frac',0),3), 'traffic', rl.get('traffic'), 'cpu', cb.get('value'), (d.get('f32') or {}).get('ms_per_step'))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
