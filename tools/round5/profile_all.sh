#!/bin/bash
# rocprofv3 evidence for every BASELINE configuration that fits one GPU (tools/profile.sh: kernel stats + FETCH / WRITE / SQ passes),
# a replay timeline each, then the driver-protocol bench lines.  Afterwards, in the build container:
#   python tools/stamp_profile.py r05_c3 ; python tools/stamp_profile.py r05_c4 egopack_oscc_B64_T32_H1024_Hp1024_bf16 ; ...
set -u
mkdir -p gpurun_out
bash tools/profile.sh r05_c3 > gpurun_out/prof_r05_c3.log 2>&1
bash tools/profile.sh r05_c2 --workload ar > gpurun_out/prof_r05_c2.log 2>&1
bash tools/profile.sh r05_c4 --workload egopack_oscc > gpurun_out/prof_r05_c4.log 2>&1
bash tools/profile.sh r05_c5 --workload mtl4 --T 256 --batch 16 > gpurun_out/prof_r05_c5.log 2>&1
for c in "c3" "c2 --workload ar" "c4 --workload egopack_oscc" "c5 --workload mtl4 --T 256 --batch 16"; do
  set -- $c; tag=$1; shift
  bash tools/timeline.sh r05_$tag "$@" > gpurun_out/tl_r05_$tag.log 2>&1
done
bash tools/round5/final_lines.sh
# (late round 5) the Hp = 4096 step's timeline, the contraction table of that width, the unprofiled phase stamps of config 4
bash tools/timeline.sh r05_hp4096 --trn-hidden 4096 > gpurun_out/tl_r05_hp4096.log 2>&1
python3 tools/gemm_bench.py --hp 4096 --variants 1,3,5,6,7,13,14,15 > gpurun_out/r05_hp4096_gemm_policy.txt 2>&1
python3 bench.py --workload egopack_oscc --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg --stamps > gpurun_out/r05_c4_stamps.json 2> gpurun_out/r05_c4_stamps.txt
