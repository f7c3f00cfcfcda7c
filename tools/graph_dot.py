#!/usr/bin/env python3
"""Read the DOT file the HIP runtime writes for an instantiated graph (DEBUG_HIP_GRAPH_DOT_PRINT=1: graph_<pid>_dot_print_<n> in the
working directory): per node the kernel, the stream the runtime assigned and its predecessors.
Usage: python3 tools/graph_dot.py FILE [--all]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"graph_1_node_(\d+)"\[[^\]]*?label="\d+\n([^\n]*)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', txt):
    nodes[int(m.group(1))] = [m.group(2), int(m.group(3)), m.group(4) == "true"]
edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_1_node_(\d+)"\s*->\s*"graph_1_node_(\d+)"', txt)]
pred, succ = {}, {}
for a, b in edges:
    pred.setdefault(b, []).append(a)
    succ.setdefault(a, []).append(b)
names = sorted({v[0] for v in nodes.values()})
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    short = {n: re.sub(r"\(.*", "", d).replace("egk::", "")[:48] for n, d in zip(names, dem)}
except Exception:
    short = {n: n[:48] for n in names}
print(f"{len(nodes)} nodes, {len(edges)} edges, streams {sorted({v[1] for v in nodes.values()})}")
for i in sorted(nodes):
    k, s, sig = nodes[i]
    p = pred.get(i, [])
    cross = [q for q in p if nodes[q][1] != s]
    if "--all" in sys.argv or cross or len(p) != 1 or len(succ.get(i, [])) != 1:
        print(f"{i:4d} s{s} {'S' if sig else ' '} {short[k]:48s} <- {p}  -> {succ.get(i, [])}")
