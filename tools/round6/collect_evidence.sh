#!/bin/bash
# In the build container, after `gpurun -- bash tools/round6/profile_all.sh`: copy the run's summaries into profiles/ (tracked),
# stamped with the commit of the tree that was measured.
set -u
cd "$(dirname "$0")/../.."
python tools/stamp_profile.py r06_c3
python tools/stamp_profile.py r06_c2 ar_B64_T32_H1024_Hp1024_bf16
python tools/stamp_profile.py r06_c4 egopack_oscc_B64_T32_H1024_Hp1024_bf16
python tools/stamp_profile.py r06_c5 mtl4_B16_T256_H1024_Hp1024_bf16
for c in c2 c3 c4 c5 hp4096; do
  cp gpurun_out/tl_r06_$c/replay.txt profiles/r06_${c}_replay_timeline.txt
  cp gpurun_out/tl_r06_$c/timeline.txt profiles/r06_${c}_timeline_summary.txt
done
for f in c2 c3 c4 c5 hp4096_mtl dry8_auto dry8_staged dry8_sharded; do cp gpurun_out/bench_r06_$f.json profiles/r06_bench_$f.json; done
grep -v "amdgpu.ids" gpurun_out/r06_r192_bench.txt > profiles/r06_r192_bench.txt
cp gpurun_out/r06_xcd_affinity.txt profiles/r06_xcd_affinity.txt
cp gpurun_out/r06_gemm_phases.txt profiles/r06_gemm_phases.txt
tail -1 gpurun_out/r06_two_rank_check.json > profiles/r06_two_rank_check.json
cp gpurun_out/r06_gpu_tests.log profiles/r06_gpu_tests.log
for f in r06_main_temporal_live.log r06_main_egopack_live.log r06_loop_gap.txt r06_loop_gap_egopack.txt r06_window_cand.txt r06_entry_loops.txt; do cp gpurun_out/$f profiles/$f; done
git status --short profiles | head -40
