for v in "" "x3_stats_split" "x3_tee" "x3_stats_split,x3_tee"; do
  echo "EGK_DISABLE=$v"; EGK_DISABLE=$v python3 -m pytest tests/test_gpu_blockwise.py -x -q -m gpu -p no:cacheprovider -k oscc_head 2>&1 | grep -E "passed|failed|d_features" | cut -c1-200
done
grep -n "OSCC head" gpurun_out/blockwise_parity.jsonl | tail -3 | cut -c1-400
