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
