#!/bin/bash
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --stamps"
timeout 300 $B --exchange-dry-run 8 > gpurun_out/stamps_dry8.json 2> gpurun_out/stamps_dry8.err; grep stamp gpurun_out/stamps_dry8.err
timeout 300 $B > gpurun_out/stamps_one.json 2> gpurun_out/stamps_one.err; grep stamp gpurun_out/stamps_one.err
