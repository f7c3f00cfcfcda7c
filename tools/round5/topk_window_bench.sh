cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tw -o tw -- python3 tools/round5/topk_window_bench.py > gpurun_out/tw.log 2>&1
cat gpurun_out/tw.log | grep "N="
python3 - <<'PY'
import csv,glob
fs=glob.glob('gpurun_out/tw/**/*kernel_trace*.csv',recursive=True); print(fs or glob.glob('gpurun_out/tw/**/*',recursive=True)[:20]); f=fs[0]
rows=[r for r in csv.DictReader(open(f)) if 'topk_window' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
# 33 calls per config
for i in range(0,len(d),33):
    seg=d[i+3:i+33]
    if seg: print('config',i//33,'topk_window avg us',sum(seg)/len(seg),'min',min(seg))
PY
rm -rf gpurun_out/tw
