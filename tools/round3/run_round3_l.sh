#!/bin/bash
# the entry point itself with live loaders at the benchmark's shapes (VERDICT r2 #6 acceptance: >= 80 % of the bench rate)
mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
timeout 900 python main_temporal.py $G $S k=1 batch_size=64 synthetic_samples=16640 synthetic_val_samples=64 num_epochs=1 enabled_tasks=[ar,lta,pnr] \
  model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 save_model=False checkpoint_dir=/tmp/ck > gpurun_out/main_temporal_live.log 2>&1
grep -n "steady state\|replayed\|iterations\|Error\|error" gpurun_out/main_temporal_live.log | tail -n 8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'])"
