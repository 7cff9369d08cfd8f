#!/usr/bin/env python3
"""Timeline of a `rocprofv3 --marker-trace --kernel-trace --output-format csv` run of the host driver: the roctx ranges
(query / batch / launch set / top-K, cudasw4_amd/csrc/host/trace_ranges.hpp) with the DP kernels that ran inside them.
    python tools/marker_summary.py <output dir>"""
import csv
import glob
import os
import sys

root = sys.argv[1]
mfiles = glob.glob(os.path.join(root, "**", "*marker_api_trace.csv"), recursive=True)
kfiles = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
if not mfiles:
    print("no marker trace under", root)
    sys.exit(1)
marks = []
for f in mfiles:
    for r in csv.DictReader(open(f)):
        name = r.get("Function") or r.get("Name") or ""
        msg = r.get("Message") or r.get("Args") or ""
        b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        marks.append((b, e, name, msg))
kernels = []
for f in kfiles:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = n[:n.index("(")] if "(" in n else n
        kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.replace("void ", "").replace("swk::", "")))
marks.sort()
kernels.sort()
t0 = min([m[0] for m in marks] + [k[0] for k in kernels[:1]])
print("%d ranges, %d kernel dispatches; times in ms since the first of them (host clock domain of the trace)" % (len(marks), len(kernels)))
print("%-9s %-9s %-9s  %s" % ("begin", "end", "ms", "range"))
stack = []
for b, e, name, msg in marks:
    while stack and stack[-1] <= b:
        stack.pop()
    depth = len(stack)
    stack.append(e)
    label = msg if msg else name
    inside = [k for k in kernels if k[0] >= b and k[0] < e and ("sw_scan" in k[2] or "sw_rows" in k[2] or "topk" in k[2])]
    extra = ""
    if inside and depth >= 1:
        names = {}
        for k in inside:
            names[k[2][:48]] = names.get(k[2][:48], 0) + 1
        extra = "   [dispatched inside: " + ", ".join("%s x%d" % kv for kv in sorted(names.items())[:4]) + "]"
    print("%9.3f %9.3f %9.3f  %s%s%s" % ((b - t0) / 1e6, (e - t0) / 1e6, (e - b) / 1e6, "  " * depth, label[:110], extra[:200]))
