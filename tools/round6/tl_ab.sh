#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/timeline.sh r6_new > /dev/null 2>&1
EGK_LIB_PATH=$PWD/tools/exp/build/libegopack_prev.so bash tools/timeline.sh r6_prev > /dev/null 2>&1
for t in new prev; do echo "== $t"; head -6 gpurun_out/tl_r6_$t/timeline.txt; done
