mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "f32_pipelined or gemm" 2>&1 | tail -8 > gpurun_out/t_f32.log; tail -n 5 gpurun_out/t_f32.log
python tools/gemm_f32_bench.py > gpurun_out/gemm_f32_bench.log 2>&1; cat gpurun_out/gemm_f32_bench.log
rm -f gpurun_out/config_parity.jsonl gpurun_out/blockwise_parity.jsonl
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/t_all.log; tail -n 6 gpurun_out/t_all.log
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/b_mtl.json 2> gpurun_out/b_mtl.err
python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['f32']['ms_per_step'], d['f32']['roofline']['frac'], d['f32']['roofline']['kernel'])" gpurun_out/b_mtl.json
