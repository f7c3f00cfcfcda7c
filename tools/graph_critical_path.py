#!/usr/bin/env python3
"""Critical path of ONE replay of a captured step: the runtime's DOT file of the instantiated graph (DEBUG_HIP_GRAPH_DOT_PRINT=1;
nodes, edges, the stream the runtime gave each node) joined with a rocprofv3 --kernel-trace CSV of the same run (start / end of
every dispatch).  Nodes of one runtime stream are dispatched in node order on one hardware queue, so the two are matched stream
by stream.  From the replay's last kernel back: the predecessor (graph edge, or the previous dispatch on the same queue) that
ended last is the one the node waited for.  Prints the path by segment (runs on one queue), the time in kernels and in gaps, and
for every node how long after its last input it started ("late": host enqueue or queue hand-off).
Usage: python3 tools/graph_critical_path.py DOT TRACE.csv [replay index from the end, default 3]"""
import csv
import re
import sys
from collections import defaultdict

dot, trace = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
txt = open(dot).read()
nodes = {}
for m in re.finditer(r'"graph_1_node_(\d+)"\[[^\]]*?label="\d+\n([^\n]*)\nStreamId:(\d+)', txt):
    nodes[int(m.group(1))] = (m.group(2), int(m.group(3)))
edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_1_node_(\d+)"\s*->\s*"graph_1_node_(\d+)"', txt)]
pred = defaultdict(list)
for a, b in edges:
    pred[b].append(a)

rows = list(csv.DictReader(open(trace)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[3]]
ends = [i for i in adam if not any("adam_kernel" in ev[j][3] for j in range(i + 1, min(i + 4, len(ev))))]
# a replay may end with a trailing launch (the RNG offset advance): take every dispatch up to the next replay's first
hi = ends[-back]
lo = ends[-back - 1] + 1
step = ev[lo:hi + 1]
while len(step) < len(nodes) and hi + 1 < len(ev):
    hi += 1
    step = ev[lo:hi + 1]
if len(step) != len(nodes):
    lo2 = lo
    # the previous replay's trailing launches belong to it, not to this one
    while len(step) > len(nodes):
        lo2 += 1
        step = ev[lo2:hi + 1]
print(f"{len(nodes)} nodes, {len(step)} dispatches in the chosen replay")
def key(name):  # the kernel's identifier out of its mangled name: _ZN3egk16gemm_pipe_kernelI... -> gemm_pipe_kernel
    i, last = (3 if name.startswith("_ZN") else 2 if name.startswith("_Z") else 0), name
    while i < len(name) and name[i].isdigit():
        j = i
        while name[j].isdigit():
            j += 1
        n = int(name[i:j])
        last = name[j:j + n]
        i = j + n
    return last


# Dispatch_Id is the order in which the host enqueued the launches: the runtime walks the nodes in creation (= id) order
disp = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])): int(r["Dispatch_Id"]) for r in rows}
ids = sorted(nodes)
allev = sorted(ev, key=lambda e: disp[(e[0], e[1])])
first = min(disp[(e[0], e[1])] for e in step)
pos0 = next(k for k, e in enumerate(allev) if disp[(e[0], e[1])] == first)
best = None
for d in range(-8, 9):  # (the replay's trailing launches behind Adam belong to it: slide the window to the graph's node list)
    cand = allev[pos0 + d: pos0 + d + len(ids)]
    if len(cand) == len(ids):
        a = sum(1 for i, e in zip(ids, cand) if key(nodes[i][0]) in e[3])
        if best is None or a > best[0]:
            best = (a, cand)
agree, step = best
print(f"dispatch k = node k: {agree} of {len(ids)} kernel names agree")
when = {i: (e[0], e[1], e[2], e[3]) for i, e in zip(ids, step)}
bys = defaultdict(list)
for i in ids:
    bys[when[i][2]].append(i)
t0 = min(w[0] for w in when.values())
qprev = {}
for s, ids in bys.items():
    for a, b in zip(ids, ids[1:]):
        qprev[b] = a
last = max(when, key=lambda i: when[i][1])
path, v = [], last
while v is not None:
    cands = list(pred.get(v, []))
    if v in qprev:
        cands.append(qprev[v])
    binding = max(cands, key=lambda p: when[p][1]) if cands else None
    ready = when[binding][1] if binding is not None else t0
    path.append((v, binding, (when[v][0] - ready) / 1e3))
    v = binding
path.reverse()
kern = sum((when[v][1] - when[v][0]) for v, _, _ in path) / 1e3
gaps = sum(max(g, 0) for _, _, g in path)
print(f"replay {(when[last][1] - t0) / 1e3:.1f} us; critical path {len(path)} nodes: {kern:.1f} us in kernels, {gaps:.1f} us in gaps")
short = lambda n: re.sub(r"^void ", "", n).replace("egk::", "")[:44]
seg_q, seg_start, seg_k, seg_g, seg_n = None, 0, 0.0, 0.0, 0
out = []
for v, b, g in path:
    q = when[v][2]
    if q != seg_q:
        if seg_q is not None:
            out.append((seg_q, seg_n, seg_start, seg_k, seg_g))
        seg_q, seg_start, seg_k, seg_g, seg_n = q, (when[v][0] - t0) / 1e3, 0.0, 0.0, 0
    seg_k += (when[v][1] - when[v][0]) / 1e3
    seg_g += max(g, 0)
    seg_n += 1
out.append((seg_q, seg_n, seg_start, seg_k, seg_g))
print("path by queue run: (queue, nodes, starts at us, us in kernels, us in gaps)")
for r in out:
    print(f"   q{r[0]} {r[1]:3d} nodes from {r[2]:8.1f}: kernels {r[3]:7.1f} gaps {r[4]:6.1f}")
big = sorted(path, key=lambda t: -t[2])[:12]
print("largest gaps on the path (node, waited-for node, gap us, kernel):")
for v, b, g in big:
    print(f"   {v:4d} <- {b}  {g:6.1f}  at {(when[v][0] - t0) / 1e3:8.1f}  {short(when[v][3])}")
# lateness of every node against its graph inputs only (not the queue predecessor): what the host / the queue adds
late = []
for v in when:
    ps = pred.get(v, [])
    ready = max((when[p][1] for p in ps), default=t0)
    qp = qprev.get(v)
    qready = when[qp][1] if qp is not None else t0
    late.append(((when[v][0] - max(ready, qready)) / 1e3, v))
late.sort(reverse=True)
print("nodes that started longest after BOTH their inputs and their queue predecessor had ended (host enqueue / hand-off):")
for g, v in late[:12]:
    print(f"   {v:4d} {g:6.1f} us at {(when[v][0] - t0) / 1e3:8.1f} q{when[v][2]} {short(when[v][3])}")
