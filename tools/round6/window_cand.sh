#!/bin/bash
# candidates per row inside the proven window of the nearest-prototype search: the live EgoPack loop's own banks / features against the bench line's
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=16384 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck64 dataset_recognition.num_class_labels=[64,64] dataset_lta.num_class_labels=[64,64]"
python main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True > /dev/null 2>&1
EGK_DBG=window_cand python main_egopack.py $G $S $C num_epochs=1 enabled_tasks=[oscc] enable_graphone=True resume_from=/tmp/ck64/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False 2>&1 | grep "window_cand" | head -4 > gpurun_out/r06_window_cand.txt
echo "--- bench line (random features and banks)" >> gpurun_out/r06_window_cand.txt
EGK_DBG=window_cand python bench.py --workload egopack_oscc --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-roofline 2>&1 | grep "window_cand" | head -3 >> gpurun_out/r06_window_cand.txt
cat gpurun_out/r06_window_cand.txt
