for r in 1 2; do
for k in 1 11 3 8; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg --min-timed-s 0 --gemm-knob $k > gpurun_out/b_k$k.json 2>/dev/null
python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['ms_per_step'],4))" gpurun_out/b_k$k.json knob$k
done; done
