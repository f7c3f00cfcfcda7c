#!/bin/bash
P="python3 tools/round4/branch_start_probe.py"
echo "== default"; $P
echo "== swap (B created first)"; $P --swap
echo "== B on the capture stream"; $P --origin-b
echo "== root launch first"; $P --root
echo "== root + swap"; $P --root --swap
echo "== long A links (100 x 6us) vs B"; $P --a 100x400000 --b 40x400000
for kv in DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=8 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 DEBUG_CLR_MAX_BATCH_SIZE=1 DEBUG_HIP_DYNAMIC_QUEUES=1 DEBUG_HIP_DYNAMIC_QUEUES=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 DEBUG_HIP_FORCE_ASYNC_QUEUE=1 ROC_ACTIVE_WAIT_TIMEOUT=100 GPU_STREAMOPS_CP_WAIT=1 GPU_STREAMOPS_CP_WAIT=0 ROC_CPU_WAIT_FOR_SIGNAL=0; do
  echo "== $kv"; env $kv $P
done
