cd $GRAFT_REPO_ROOT
run() { (cd $1 && python bench.py $2 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'])"); }
for r in 1 2; do run . "--workload mtl4 --T 256 --batch 16"; run _old "--workload mtl4 --T 256 --batch 16"; done
for r in 1 2; do run . ""; run _old ""; done
