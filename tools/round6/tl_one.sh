#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline.sh r6_$1 > /dev/null 2>&1
head -2 gpurun_out/tl_r6_$1/timeline.txt
grep -E "rowln|graphln|csr_gather|pe_add" gpurun_out/tl_r6_$1/replay.txt | awk '{print $2, $4}' | cut -c1-60
