set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_precise.py -x -q 2>&1 | tail -15 > gpurun_out/t_precise.log
python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -15 > gpurun_out/t_configs.log
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_configs.py --deselect tests/test_gpu_precise.py 2>&1 | tail -15 > gpurun_out/t_all.log
python bench.py --workload egopack_oscc --steps 50 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline > gpurun_out/b_ego_precise.json 2> gpurun_out/b_ego_precise.err
EGK_DISABLE=precise_search python bench.py --workload egopack_oscc --steps 50 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline > gpurun_out/b_ego_plain.json 2> gpurun_out/b_ego_plain.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/b_mtl.json 2> gpurun_out/b_mtl.err
tail -3 gpurun_out/t_precise.log gpurun_out/t_configs.log gpurun_out/t_all.log
for f in gpurun_out/b_ego_precise.json gpurun_out/b_ego_plain.json gpurun_out/b_mtl.json; do python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d[\"ms_per_step\"],4), d.get(\"timed_blocks\"), d.get(\"block_ms_min\"), d.get(\"block_ms_max\"), (d.get(\"roofline\") or {}).get(\"frac\"), ((d.get(\"roofline\") or {}).get(\"replay_dominant\") or {}).get(\"kernel\"), (d.get(\"f32\") or {}).get(\"ms_per_step\"))" $f; done
