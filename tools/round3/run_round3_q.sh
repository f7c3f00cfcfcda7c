#!/bin/bash
mkdir -p gpurun_out/prof_f32
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f32 -o bench -- python3 bench.py --compute f32 --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/prof_f32/log.txt 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_f32/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%-90s %6s %10.1f %8.2f %5s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/prof_f32 -name "*kernel_trace.csv" -delete
