#!/usr/bin/env python3
"""Timeline of ONE graph replay from a rocprofv3 --kernel-trace CSV: which queues run what, how much of the step the
device is idle, and the gaps between consecutive kernels on the busiest queue (launch-to-launch dependency latency).
Usage: python tools/trace_timeline.py <bench_kernel_trace.csv> [replay index from the end, default 3]"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]) for r in rows]
ev.sort()
# replays are delimited by the Adam kernel (one per step, the last kernel of a replay)
# (a captured step may issue Adam in several launches: early slices beside the last weight gradient, then the rest --
#  the step ends with the last Adam launch of such a group)
#  the step ends with the last Adam launch of such a group; a slice may also run in the MIDDLE of backward (GraphONE's), so the
#  launches per step are counted against the one hyper-parameter launch every step has, and the ends are taken from the back)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[3]]
hyper = sum(1 for e in ev if "adam_hyper_kernel" in e[3])
per = max(1, round(len(adam) / hyper)) if hyper else 1
ends = adam[::-1][::per][::-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hi = ends[-back]
lo = ends[-back - 1] + 1
step = ev[lo: hi + 1]
t0, t1 = step[0][0], max(e[1] for e in step)
print(f"replay of {len(step)} kernels, {(t1 - t0) / 1e3:.1f} us from first start to last end")
# union of busy intervals
iv = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"device busy (union over queues) {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us; sum of kernel durations {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us")
byq = defaultdict(list)
for e in step:
    byq[e[2]].append(e)
for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    dur = sum(e[1] - e[0] for e in es)
    gaps = [es[i + 1][0] - es[i][1] for i in range(len(es) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"queue {q}: {len(es):3d} kernels, busy {dur / 1e3:7.1f} us; gaps between consecutive kernels: {len(pos)} positive, "
          f"median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us, total {sum(pos) / 1e3:.1f} us")
# concurrency histogram
pts = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
level, last, hist = 0, pts[0][0], defaultdict(int)
for t, d in pts:
    hist[level] += t - last
    level += d
    last = t
print("time with k kernels in flight: " + ", ".join(f"{k}: {v / 1e3:.0f} us" for k, v in sorted(hist.items())))
if len(sys.argv) > 3:  # full listing of the replay: start (us from the replay's first start), duration, queue, kernel
    import re
    with open(sys.argv[3], "w") as f:
        for s, e, q, k in step:
            k = re.sub(r"^void ", "", k).replace("egk::", "")
            f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {k[:100]}\n")
