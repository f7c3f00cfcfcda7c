#!/usr/bin/env python3
"""stdin: the JSON line of bench.py -> one short line (tag ms/step value)."""
import json
import sys

line = sys.stdin.read().strip()
try:
    d = json.loads(line)
    print(f"[{sys.argv[1] if len(sys.argv) > 1 else ''}] {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} {d['unit']}")
except Exception:
    print(f"[{sys.argv[1] if len(sys.argv) > 1 else ''}] FAIL {line[-300:]}")
