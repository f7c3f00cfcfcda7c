cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --workload egopack_oscc --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for r in 1 2; do run X=1; run EGK_TRAIN_AFTER=fwd_trn_done; run "EGK_TRAIN_AFTER=fwd_sage[0]"; run "EGK_TRAIN_AFTER=fwd_sage[1]"; run "EGK_TRAIN_AFTER=fwd_sage[2]"; done
