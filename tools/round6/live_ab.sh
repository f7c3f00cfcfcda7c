#!/bin/bash
# same-box alternation of the temporal loop with a switch off / on:  bash tools/round6/live_ab.sh <switch> [rounds]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=16384 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
for r in $(seq ${2:-3}); do
  for off in "" "$1"; do
    echo -n "EGK_DISABLE='$off' "; EGK_DISABLE=$off timeout 900 python main_temporal.py $G $S $C num_epochs=2 enabled_tasks=[ar,lta,pnr] save_model=False 2>&1 | grep "steady" | sed 's/.*steady state \([0-9.]*\) ms.*/\1/' | tr '\n' ' '; echo
  done
done
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'])"
