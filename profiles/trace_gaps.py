#!/usr/bin/env python
"""Where the device time of a run goes, from a rocprofv3 --kernel-trace CSV: over the last `frac` of the dispatches, the
span, the union of busy intervals (any queue), the idle remainder, and per kernel: launches, total and mean duration.
    python profiles/trace_gaps.py <dir with *_kernel_trace.csv> [frac=0.4]"""
import collections
import csv
import os
import sys


def find(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


path = find(sys.argv[1], "kernel_trace.csv")
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "")
    n = n[:n.find("(")] if "(" in n else n
    tot[n[:70]][0] += 1
    tot[n[:70]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"{len(rows)} dispatches, span {span / 1e6:.3f} ms, device busy (union) {busy / 1e6:.3f} ms, idle {100 * (span - busy) / span:.1f} %, "
      f"sum of durations {sum(v[1] for v in tot.values()) / 1e6:.3f} ms")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {t / 1e6:9.3f} ms  {c:6d} x {t / c / 1e3:9.1f} us  {n}")
