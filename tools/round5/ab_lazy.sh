set -x
cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --workload egopack_oscc --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LAZY', d['ms_per_step'])"
EGK_DISABLE=x3_lazy_input python bench.py --workload egopack_oscc --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EAGER', d['ms_per_step'])"
done
python -m pytest tests/test_gpu_configs.py tests/test_gpu_precise.py tests/test_gpu_blockwise.py -x -q 2>&1 | tail -5
python -m pytest tests/test_gpu_models.py -x -q -k "egopack or graphone or precise" 2>&1 | tail -3
