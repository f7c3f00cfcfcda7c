#!/bin/bash
# every opt-in switch still runs (one short bench of configs 3 and 4 each)
run() { name=$1; shift; "$@" 2>gpurun_out/optin_err.txt | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-roofline"
for sw in gather_fusion splitk_in_launch segmented_replay=4 segmented_replay=2,plan_event_nodes; do
  EGK_ENABLE=$sw run c3_$sw python3 bench.py $C
  EGK_ENABLE=$sw run c4_$sw python3 bench.py $C --workload egopack_oscc
done
for sw in precise_search fork_order tail_group proj_park heads_flush banks_ride wg4 ln_fusion rowdot_head banded_gather x3_tee; do
  EGK_DISABLE=$sw run c3_no_$sw python3 bench.py $C
done
for sw in precise_search x3_tee x3_stats_split x3_grouped_aux oscc_one_pass wgrad_grouping deferred_forks; do
  EGK_DISABLE=$sw run c4_no_$sw python3 bench.py $C --workload egopack_oscc
done
run c3_base python3 bench.py $C
run c4_base python3 bench.py $C --workload egopack_oscc
EGK_DISABLE=oscc_one_pass run c5_no_oscc_one_pass python3 bench.py $C --workload mtl4 --T 256 --batch 16
run c5_base python3 bench.py $C --workload mtl4 --T 256 --batch 16
