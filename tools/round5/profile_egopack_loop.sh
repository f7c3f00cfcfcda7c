#!/bin/bash
# cProfile of main_egopack.py's epoch at the benchmark's shapes (after tools/round5/entry_loops.sh wrote /tmp/ck on the same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=8192 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
timeout 900 python main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True > gpurun_out/p1.log 2>&1
timeout 900 python -m cProfile -o /tmp/egopack.prof main_egopack.py $G $S $C num_epochs=1 enabled_tasks=[oscc] enable_graphone=True resume_from=/tmp/ck/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False > gpurun_out/p2.log 2>&1
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/egopack.prof')
p.sort_stats('cumtime').print_stats('engine.py|data.py|main_egopack|graphs.py|feature_store|ops.py', 45)
PY
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/egopack.prof')
p.print_callers('custom_ops.py:50')
p.print_callers('graphs.py:141')
p.sort_stats('tottime').print_stats(12)
PY
