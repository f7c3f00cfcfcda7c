#!/bin/bash
# Runtime-configuration A/B of the headline step (same box, alternating): kernel-argument placement and the graph replay path.
mkdir -p gpurun_out/envab
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.5"
run() {  # name, env...
  name=$1; shift
  env "$@" $B > gpurun_out/envab/$name.json 2> gpurun_out/envab/$name.err
  python3 - "$name" <<'P'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/envab/{n}.json").read().strip().splitlines()[-1])
    print(n, "ms_per_step", round(d["ms_per_step"],4), "min", round(d["block_ms_min"],4), "max", round(d["block_ms_max"],4))
except Exception as e:
    print(n, "FAILED", e)
P
}
for rep in 1 2; do
  run base_$rep X=1
  run devkernarg1_$rep HIP_FORCE_DEV_KERNARG=1
  run devkernarg0_$rep HIP_FORCE_DEV_KERNARG=0
  run pktcap1_$rep DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run pktcap0_$rep DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run hwq2_$rep GPU_MAX_HW_QUEUES=2
  run hwq8_$rep GPU_MAX_HW_QUEUES=8
done 2>&1 | tee gpurun_out/envab/summary.txt
