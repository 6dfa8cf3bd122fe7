#!/usr/bin/env python
"""The idle gaps (no kernel on ANY queue) of the last `nsteps` training steps of a rocprofv3 --kernel-trace CSV of a DeepFM
run, grouped by the kernel that ends before the gap and the one that starts after it: where the device waits.
    python profiles/trace_idle.py <dir with *_kernel_trace.csv> [nsteps=16] [skip_last=4]"""
import collections
import csv
import os
import sys

path = None
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("kernel_trace.csv"):
            path = os.path.join(root, f)
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def name(r):
    n = r["Kernel_Name"].replace("void ", "")
    n = n[:n.find("(")] if "(" in n else n
    n = n[:n.find("<")] if "<" in n else n
    return n[-40:]


idx = [i for i, r in enumerate(rows) if "lazy_adam_catchup" in r["Kernel_Name"]]
a, b = idx[-(nsteps + skip) - 1], idx[-skip - 1]
sel = rows[a:b]
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"{nsteps} steps, {span / nsteps / 1e3:.1f} us per step")
ev = sorted(sel, key=lambda r: int(r["Start_Timestamp"]))
gaps = collections.defaultdict(lambda: [0, 0])
cur_e, cur_r = int(ev[0]["End_Timestamp"]), ev[0]
for r in ev[1:] + [rows[b]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_e:
        g = gaps[(name(cur_r), name(r))]
        g[0] += 1
        g[1] += s - cur_e
    if e > cur_e:
        cur_e, cur_r = e, r
tot = sum(v[1] for v in gaps.values())
print(f"idle (no kernel on any queue): {tot / nsteps / 1e3:.1f} us per step")
for (p, q), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / nsteps / 1e3:7.1f} us/step  {c / nsteps:5.2f} x {t / c / 1e3:6.1f} us   after {p}  ->  before {q}")
