#!/bin/bash
# In the build container, after `gpurun -- bash tools/round5/profile_all.sh`: copy the run's summaries into profiles/ (tracked),
# stamped with the commit of the tree that was measured.
set -u
cd "$(dirname "$0")/../.."
python tools/stamp_profile.py r05_c3
python tools/stamp_profile.py r05_c2 ar_B64_T32_H1024_Hp1024_bf16
python tools/stamp_profile.py r05_c4 egopack_oscc_B64_T32_H1024_Hp1024_bf16
python tools/stamp_profile.py r05_c5 mtl4_B16_T256_H1024_Hp1024_bf16
for c in c2 c3 c4 c5 hp4096; do
  cp gpurun_out/tl_r05_$c/replay.txt profiles/r05_${c}_replay_timeline.txt
  cp gpurun_out/tl_r05_$c/timeline.txt profiles/r05_${c}_timeline_summary.txt
done
for f in c2 c3 c4 c5 hp4096_mtl dry8_staged; do cp gpurun_out/bench_r05_$f.json profiles/r05_bench_$f.json; done
grep -v "amdgpu.ids" gpurun_out/r05_hp4096_gemm_policy.txt > profiles/r05_hp4096_gemm_policy.txt
grep "\[stamp\]" gpurun_out/r05_c4_stamps.txt | grep -v "\[[3-9]\]\|\[1[0-9]\]" > profiles/r05_c4_phase_stamps.txt
tail -1 gpurun_out/r05_c4_stamps.json | cut -c1-400 >> profiles/r05_c4_phase_stamps.txt
git status --short profiles | head -40
