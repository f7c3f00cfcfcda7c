mkdir -p gpurun_out
python tools/gemm_group_pack_bench.py 2>&1 | tail -8
python -m pytest tests/test_gpu_kernels.py -x -q -k "grouped or group" 2>&1 | tail -3
for r in 1 2 3; do
for k in 300 301; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-f32-leg --min-timed-s 0 --gemm-knob $k > gpurun_out/b_k$k.json 2>/dev/null
python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['ms_per_step'],4))" gpurun_out/b_k$k.json knob$k
done; done
