mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t_all.log
for tag in "" "precise_stream" "precise_search"; do
EGK_DISABLE=$tag python bench.py --workload egopack_oscc --steps 50 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline > gpurun_out/b_ego_$tag.json 2> gpurun_out/b_ego_$tag.err
python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['ms_per_step'],4), d.get('timed_blocks'), d.get('block_ms_min'), d.get('block_ms_max'))" gpurun_out/b_ego_$tag.json
done
tail -n 4 gpurun_out/t_all.log
