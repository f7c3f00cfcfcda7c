cd $GRAFT_REPO_ROOT
run() { env $2 python bench.py --trn-hidden 4096 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg --egk-tune 5=$1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zero blocks=$1 $2', d['ms_per_step'])"; }
for r in 1 2; do run 64 X=1; run 16 X=1; run 256 X=1; run 1024 X=1; run 64 EGK_DISABLE=zero_stream; done
