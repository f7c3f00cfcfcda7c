#!/bin/bash
# The reference's two entry points at the benchmark's shapes with live loaders on the device-resident store (as tools/round3/run_round3_r.sh):
# their steady-state step rate next to the bench lines of the same box.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=32768 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
timeout 900 python main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True > gpurun_out/r06_main_temporal_live.log 2>&1
grep -n "steady state" gpurun_out/r06_main_temporal_live.log | tail -n 2
# (64 x 64 action classes and 32768 samples: ~4095 distinct (verb, noun) pairs seen = the bench line's K = 4096 prototypes per bank;
#  the default 115 x 478 classes give ~7600 prototypes from 8192 samples and a search twice as long)
K="dataset_recognition.num_class_labels=[64,64] dataset_lta.num_class_labels=[64,64]"
timeout 900 python main_temporal.py $G $S $C $K num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True checkpoint_dir=/tmp/ck64 > /dev/null 2>&1
timeout 900 python main_egopack.py $G $S $C $K checkpoint_dir=/tmp/ck64 num_epochs=1 enabled_tasks=[oscc] enable_graphone=True resume_from=/tmp/ck64/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False > gpurun_out/r06_main_egopack_live.log 2>&1
grep -n "steady state\|prototype banks" gpurun_out/r06_main_egopack_live.log | tail -n 3
for w in "" "--workload egopack_oscc"; do python bench.py $w --steps 50 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', '$w', d['ms_per_step'], d['value'])"; done
# where the host's time goes in the temporal loop (cProfile, sorted by own time)
timeout 900 python -m cProfile -o /tmp/mt.prof main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=False > /dev/null 2>&1
python -c "import pstats; pstats.Stats('/tmp/mt.prof').sort_stats('tottime').print_stats(45)" > gpurun_out/r06_main_temporal_cprofile.txt 2>&1
