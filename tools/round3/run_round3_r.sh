#!/bin/bash
# main_egopack on the resident store: the entry-point test, then its loop at the benchmark's shapes (config 4: OSCC + GraphONE)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_entrypoints.py -x -q -m gpu > gpurun_out/t_entry.log 2>&1; tail -n 5 gpurun_out/t_entry.log
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=8192 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
timeout 900 python main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True > gpurun_out/egopack_phase1.log 2>&1
grep -n "steady state" gpurun_out/egopack_phase1.log | tail -n 2
timeout 900 python main_egopack.py $G $S $C num_epochs=1 enabled_tasks=[oscc] enable_graphone=True resume_from=/tmp/ck/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False > gpurun_out/main_egopack_live.log 2>&1
grep -n "steady state\|replayed\|iterations\|Error\|error" gpurun_out/main_egopack_live.log | tail -n 8; tail -n 3 gpurun_out/main_egopack_live.log | cut -c1-300
python bench.py --workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench egopack_oscc', d['ms_per_step'], d['value'])"
