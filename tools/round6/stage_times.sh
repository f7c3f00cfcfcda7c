#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=16384 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
EGK_DBG=stage_profile timeout 900 python main_temporal.py $G $S $C num_epochs=2 enabled_tasks=[ar,lta,pnr] save_model=False > gpurun_out/stage_profile.log 2>&1
grep "steady\|ring:" gpurun_out/stage_profile.log
EGK_DISABLE=staging_priority timeout 900 python main_temporal.py $G $S $C num_epochs=2 enabled_tasks=[ar,lta,pnr] save_model=False 2>&1 | grep "steady\|ring:"
