"""What the device does BETWEEN two replays of the captured step inside a live training loop (rocprofv3 kernel + memory-copy traces of an
entry point): every kernel / copy from the last optimizer launch of one step to the tenth kernel of the next, with start offsets.
Usage: python tools/round6/loop_gap.py <dir with *_kernel_trace.csv, *_memory_copy_trace.csv> [step from the end, default 5]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"q{r['Queue_Id']}", re.sub(r"^void ", "", r["Kernel_Name"]).replace("egk::", "")[:90]))
for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", f"{r.get('Direction', '')} {r.get('Bytes', r.get('Size', ''))} B"))
ev.sort()
hyper = [i for i, e in enumerate(ev) if "adam_hyper_kernel" in e[3]]
print(f"{len(ev)} events, {len(hyper)} steps")
periods = [(ev[hyper[i + 1]][0] - ev[hyper[i]][0]) / 1e3 for i in range(len(hyper) - 1)]
tail = sorted(periods[-60:])
print(f"step period over the last 60 steps: median {tail[len(tail) // 2]:.1f} us, min {tail[0]:.1f}, max {tail[-1]:.1f}")
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[3]]
h0, h1 = hyper[-back - 1], hyper[-back]
last_adam = max(i for i in adam if i < h1 and i > h0) if any(h0 < i < h1 for i in adam) else h1
# the step that starts after last_adam: print from 6 events before the last Adam launch to 14 events after it
first_next = None
t_end = ev[last_adam][1]
print(f"\\nlast optimizer launch of a step ends at t = 0")
for s, e, q, k in ev[last_adam - 6:last_adam + 30]:
    print(f"{(s - t_end) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {q:>5s} {k}")
# one whole step (from the first event behind the previous step's last optimizer launch to this step's), listed like tools/trace_timeline.py
prev_adam = max(i for i in adam if i < h0) if any(i < h0 for i in adam) else 0
if len(sys.argv) > 3:
    step = [e for e in ev[prev_adam + 1:last_adam + 1]]
    t0 = step[0][0]
    with open(sys.argv[3], "w") as f:
        for s, e, q, k in step:
            f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {q} {k}\n")
