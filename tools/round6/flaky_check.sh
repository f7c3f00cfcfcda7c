#!/bin/bash
# the loops' staging thread + the two captured steps, over and over: entry-point tests N times, then a multi-epoch live run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in $(seq ${1:-8}); do
  timeout 600 python -m pytest tests/test_gpu_entrypoints.py -x -q 2>&1 | tail -1
done
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=8200 synthetic_val_samples=256 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
timeout 1200 python main_temporal.py $G $S $C num_epochs=12 enabled_tasks=[ar,lta,pnr] save_model=True 2>&1 | grep "steady\|replayed\|train loss\|Error\|Traceback" | tail -40
