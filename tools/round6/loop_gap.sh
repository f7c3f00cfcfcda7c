#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=6144 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
if [ "${2:-mtl}" = "egopack" ]; then
C="$C dataset_recognition.num_class_labels=[64,64] dataset_lta.num_class_labels=[64,64] synthetic_samples=16384"  # (~4020 prototypes per bank: the bench line's K)
python3 main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True > /dev/null 2>&1
rm -rf /tmp/lg; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lg -o loop -- python3 main_egopack.py $G $S $C num_epochs=1 enabled_tasks=[oscc] enable_graphone=True resume_from=/tmp/ck/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False > gpurun_out/loop_gap_run.log 2>&1
grep steady gpurun_out/loop_gap_run.log
python3 tools/round6/loop_gap.py /tmp/lg ${1:-5} gpurun_out/r06_loop_step_egopack.txt > gpurun_out/r06_loop_gap_egopack.txt 2>&1; cat gpurun_out/r06_loop_gap_egopack.txt
exit 0
fi
rm -rf /tmp/lg; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lg -o loop -- python3 main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=False > gpurun_out/loop_gap_run.log 2>&1
grep steady gpurun_out/loop_gap_run.log
python3 tools/round6/loop_gap.py /tmp/lg ${1:-5} gpurun_out/r06_loop_step.txt > gpurun_out/r06_loop_gap.txt 2>&1; cat gpurun_out/r06_loop_gap.txt
