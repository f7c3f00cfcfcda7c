#!/usr/bin/env python3
"""Pretty-print a bench.py log produced with --kernel-table (table on stderr + JSON line on stdout)."""
import json
import sys

t = open(sys.argv[1]).read()
j = t.rindex('{"metric')
line = json.loads(t[j:].splitlines()[0])
try:
    i = t.index("{\n")
    tab = json.loads(t[i:j])
    tot = sum(v["ms_per_step"] for v in tab.values())
    print(f"{'kernel':22s} {'n/step':>6s} {'ms/step':>8s} {'TF/s':>7s} {'GB/s':>7s}")
    for k, v in list(tab.items())[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
        tf = f"{v['tflops']:.1f}" if v["tflops"] else ""
        gb = f"{v['gbs']:.0f}" if v["gbs"] else ""
        print(f"{k:22s} {v['launches_per_step']:6.1f} {v['ms_per_step']:8.3f} {tf:>7s} {gb:>7s}")
    print(f"{'sum (eager, event-timed)':22s} {'':6s} {tot:8.3f}")
except ValueError:
    pass
print(f"value={line['value']:.0f} {line['unit']}  ms/step={line['ms_per_step']:.3f}  dtype={line['dtype']}")
print("roofline:", json.dumps(line["roofline"]))
print("cpu_baseline:", json.dumps(line["cpu_baseline"]))
