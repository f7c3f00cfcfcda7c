#!/bin/bash
# Round 6 evidence, one GPU call: rocprofv3 kernel stats + FETCH / WRITE / SQ passes and a replay timeline for every BASELINE
# configuration that fits one GPU (tools/profile.sh, tools/timeline.sh), the driver-protocol bench lines, the 8-rank code path with no
# link time in its three modes, the contraction tables and phase stamps behind DESIGN 3.2.  Afterwards, in the build container:
#   bash tools/round6/collect_evidence.sh
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile.sh r06_c3 > gpurun_out/prof_r06_c3.log 2>&1
bash tools/profile.sh r06_c2 --workload ar > gpurun_out/prof_r06_c2.log 2>&1
bash tools/profile.sh r06_c4 --workload egopack_oscc > gpurun_out/prof_r06_c4.log 2>&1
bash tools/profile.sh r06_c5 --workload mtl4 --T 256 --batch 16 > gpurun_out/prof_r06_c5.log 2>&1
for c in "c3" "c2 --workload ar" "c4 --workload egopack_oscc" "c5 --workload mtl4 --T 256 --batch 16" "hp4096 --trn-hidden 4096"; do
  set -- $c; tag=$1; shift
  bash tools/timeline.sh r06_$tag "$@" > gpurun_out/tl_r06_$tag.log 2>&1
done
python3 bench.py > gpurun_out/bench_r06_c3.json 2> gpurun_out/bench_r06_c3.err
python3 bench.py --workload ar > gpurun_out/bench_r06_c2.json 2> gpurun_out/bench_r06_c2.err
python3 bench.py --workload egopack_oscc > gpurun_out/bench_r06_c4.json 2> gpurun_out/bench_r06_c4.err
python3 bench.py --workload mtl4 --T 256 --batch 16 > gpurun_out/bench_r06_c5.json 2> gpurun_out/bench_r06_c5.err
python3 bench.py --trn-hidden 4096 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r06_hp4096_mtl.json 2> gpurun_out/bench_r06_hp4096_mtl.err
python3 bench.py --exchange-dry-run 8 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r06_dry8_auto.json 2> gpurun_out/bench_r06_dry8_auto.err
python3 bench.py --exchange-dry-run 8 --exchange-graph staged --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r06_dry8_staged.json 2> gpurun_out/bench_r06_dry8_staged.err
EGK_ENABLE=sharded_update python3 bench.py --exchange-dry-run 8 --no-cpu-baseline --no-f32-leg > gpurun_out/bench_r06_dry8_sharded.json 2> gpurun_out/bench_r06_dry8_sharded.err
for f in c2 c3 c4 c5 hp4096_mtl dry8_auto dry8_staged dry8_sharded; do python3 - gpurun_out/bench_r06_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rl=d.get('roofline') or {}; cb=d.get('cpu_baseline') or {}
    print(sys.argv[1], round(d['ms_per_step'],4), round(d['value']), d['config'].get('capture'), rl.get('kernel'), round(rl.get('frac',0),3), 'step', round((rl.get('step') or {}).get('frac',0),3), 'traffic', rl.get('traffic'), 'cpu', cb.get('value'), (d.get('f32') or {}).get('ms_per_step'))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
python3 tools/round6/r192_bench.py --variants 8,16,1 > gpurun_out/r06_r192_bench.txt 2>&1
tools/exp/build/xcd_affinity > gpurun_out/r06_xcd_affinity.txt 2>&1
for v in 8 16; do for K in 1024 2048 4608; do python3 tools/gemm_stamps.py --phases 6144 1024 $K 0 0 1 $v 2>&1 | grep -v "amdgpu.ids"; done; done > gpurun_out/r06_gemm_phases.txt
# the live loops: steady-state rates next to the bench lines, what the device does between two replays, the prototype search's candidates
bash tools/round6/entry_loops.sh > gpurun_out/r06_entry_loops.txt 2>&1
bash tools/round6/loop_gap.sh > /dev/null 2>&1
bash tools/round6/loop_gap.sh 5 egopack > /dev/null 2>&1
bash tools/round6/window_cand.sh > /dev/null 2>&1
python3 tools/two_rank_check.py > gpurun_out/r06_two_rank_check.json 2> gpurun_out/r06_two_rank_check.err
python3 -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_tests.log 2>&1; tail -3 gpurun_out/r06_gpu_tests.log
