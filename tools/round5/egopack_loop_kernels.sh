cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
G="dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident"
S="dataset_recognition.T=32 dataset_lta.T=32 dataset_oscc.T=32 dataset_pnr.T=32 dataset_recognition.n_videos=8 dataset_lta.n_videos=8 dataset_oscc.n_videos=8 dataset_pnr.n_videos=8 dataset_recognition.frames=4000 dataset_lta.frames=4000 dataset_oscc.frames=4000 dataset_pnr.frames=4000"
C="k=1 batch_size=64 synthetic_samples=8192 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 compute=bf16 checkpoint_dir=/tmp/ck"
timeout 900 python main_temporal.py $G $S $C num_epochs=1 enabled_tasks=[ar,lta,pnr] save_model=True > gpurun_out/p1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ek -o ek -- python3 main_egopack.py $G $S $C num_epochs=1 enabled_tasks=[oscc] enable_graphone=True resume_from=/tmp/ck/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False > gpurun_out/p2.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ek/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows[:40]:
    c = int(r['Calls']); tot += c
    print(f"{c:7d} calls {c/126:7.2f}/step avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
print('total calls', sum(int(r['Calls']) for r in rows), 'per step ~', sum(int(r['Calls']) for r in rows) / 126)
PY
rm -rf gpurun_out/ek
