#!/bin/bash
# The reference's two entry points at the benchmark's shapes with live loaders on the device-resident store (as tools/round3/run_round3_r.sh):
# their steady-state step rate next to the bench lines of the same box.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=8192 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
timeout 900 python main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True > gpurun_out/r05_main_temporal_live.log 2>&1
grep -n "steady state" gpurun_out/r05_main_temporal_live.log | tail -n 2
timeout 900 python main_egopack.py $G $S $C num_epochs=1 enabled_tasks=[oscc] enable_graphone=True resume_from=/tmp/ck/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False > gpurun_out/r05_main_egopack_live.log 2>&1
grep -n "steady state\|replayed\|Error\|error" gpurun_out/r05_main_egopack_live.log | tail -n 6
for w in "" "--workload egopack_oscc"; do python bench.py $w --steps 50 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', '$w', d['ms_per_step'], d['value'])"; done
