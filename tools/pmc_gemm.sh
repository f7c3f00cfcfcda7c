#!/bin/bash
# SQ counters of one contraction shape.  Usage: bash tools/pmc_gemm.sh <tag> M N K tA tB [splitk] [pipeline]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
CTRS=${CTRS:-"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"}
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o g -- python3 tools/gemm_one.py "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, collections
from pathlib import Path
f = sorted(Path(sys.argv[1]).rglob("*counter_collection.csv"))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gemm" not in k: continue
    acc[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v:16.0f}")
PY
