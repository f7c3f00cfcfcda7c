#!/bin/bash
# rocprofv3 evidence for every BASELINE configuration that fits one GPU (tools/profile.sh: kernel stats + FETCH / WRITE / SQ passes),
# plus a replay timeline each.  Then:  python tools/stamp_profile.py r04_c3 ; python tools/stamp_profile.py r04_c4 <key> ...
set -u
bash tools/profile.sh r04_c3 > gpurun_out/prof_r04_c3.log 2>&1
bash tools/profile.sh r04_c2 --workload ar > gpurun_out/prof_r04_c2.log 2>&1
bash tools/profile.sh r04_c4 --workload egopack_oscc > gpurun_out/prof_r04_c4.log 2>&1
bash tools/profile.sh r04_c5 --workload mtl4 --T 256 --batch 16 > gpurun_out/prof_r04_c5.log 2>&1
for c in "c3" "c2 --workload ar" "c4 --workload egopack_oscc" "c5 --workload mtl4 --T 256 --batch 16"; do
  set -- $c; tag=$1; shift
  bash tools/timeline.sh r04_$tag "$@" > gpurun_out/tl_r04_$tag.log 2>&1
done
# bench lines (full protocol incl. roofline + cpu baseline where defined)
python3 bench.py > gpurun_out/bench_r04_c3.json 2> gpurun_out/bench_r04_c3.err
python3 bench.py --workload ar > gpurun_out/bench_r04_c2.json 2> gpurun_out/bench_r04_c2.err
python3 bench.py --workload egopack_oscc > gpurun_out/bench_r04_c4.json 2> gpurun_out/bench_r04_c4.err
python3 bench.py --workload mtl4 --T 256 --batch 16 > gpurun_out/bench_r04_c5.json 2> gpurun_out/bench_r04_c5.err
tail -c 600 gpurun_out/bench_r04_c*.json
