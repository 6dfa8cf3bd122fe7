#!/usr/bin/env python
"""Timeline of ONE training step out of a rocprofv3 --kernel-trace CSV of a DeepFM run: every dispatch between two
consecutive lazy_adam_catchup launches (start offset, duration, queue, kernel), and the mean step span over 16 steps.
    python profiles/trace_step.py <dir with *_kernel_trace.csv> <index of the step (catch-up launch number)>"""
import csv
import os
import sys

path = None
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("kernel_trace.csv"):
            path = os.path.join(root, f)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1050
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "lazy_adam_catchup" in r["Kernel_Name"]]
a, b = idx[k - 8], idx[k + 8]
print(f"{len(idx)} steps in the trace; steps {k - 8}..{k + 8}: {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 16e6:.4f} ms per step")
a1, b1 = idx[k], idx[k + 1]
t0 = int(rows[a1]["Start_Timestamp"])
qs = {}
busy = 0
for r in rows[a1:b1]:
    n = r["Kernel_Name"].replace("void ", "")
    n = n[:n.find("(")] if "(" in n else n
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q}  {n[:60]}")
