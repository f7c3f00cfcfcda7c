#!/bin/bash
# soak of the live-loader loop: 8 epochs x 260 steps of main_temporal at the benchmark's shapes (memory must not grow, rate must hold)
mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
timeout 1200 python main_temporal.py $G $S k=1 batch_size=64 synthetic_samples=16640 synthetic_val_samples=64 num_epochs=8 enabled_tasks=[ar,lta,pnr] \
  model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 save_model=False checkpoint_dir=/tmp/ck > gpurun_out/main_temporal_soak.log 2>&1
grep -n "steady state\|train loss" gpurun_out/main_temporal_soak.log | cut -c1-260
timeout 900 python tools/soak.py 3000 2>&1 | tail -n 3
