set -u
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg"
for cfg in "--batch 1 --T 2" "--batch 1 --T 2 --serial-heads --no-wgrad-streams --no-early-adam" "--batch 1 --T 2 --workload ar" "--batch 1 --T 2 --workload ar --no-wgrad-streams --no-early-adam" "" "--serial-heads --no-wgrad-streams --no-early-adam" "--no-wgrad-streams" "--no-early-adam"; do
  echo "== $cfg"; $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4))"
done
python tools/launch_floor.py
