#!/usr/bin/env python
"""Per training step of a rocprofv3 --kernel-trace CSV of a DeepFM run (last `nsteps` steps before the final `skip`):
period (catch-up start to next catch-up start), the durations of the three biggest kernels, when the last kernel of the
step ends, and the idle time in front of the next step's first kernel.
    python profiles/trace_periods.py <dir with *_kernel_trace.csv> [nsteps=20] [skip=4]"""
import csv
import os
import sys

path = None
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("kernel_trace.csv"):
            path = os.path.join(root, f)
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "lazy_adam_catchup" in r["Kernel_Name"]]
print("step  period  catchup  gather  grad_seg/gemm  wgrad  last_end  idle_before_next  first kernels after the optimizer")
for k in range(len(idx) - nsteps - skip - 1, len(idx) - skip - 1):
    a, b = idx[k], idx[k + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    dur = {}
    end = 0
    tail = []
    seen_adam = False
    for r in rows[a:b]:
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for key in ("lazy_adam_catchup", "embed_gather_linear", "embed_grad_gemm", "linear_wgrad"):
            if key in n or (key == "embed_grad_gemm" and "embed_grad_seg_kernel" in n):  # (round 5: the segment-sum-first launch)
                dur[key] = d
        end = max(end, int(r["End_Timestamp"]))
        if seen_adam:
            tail.append(f"{n.replace('void ', '')[:18]}@{(int(r['Start_Timestamp']) - t0) / 1e3:.0f}")
        if "adam_kernel" in n:
            seen_adam = True
    period = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    print(f"{k:5d} {period:7.1f} {dur.get('lazy_adam_catchup', 0):7.1f} {dur.get('embed_gather_linear', 0):7.1f} "
          f"{dur.get('embed_grad_gemm', 0):8.1f} {dur.get('linear_wgrad', 0):7.1f} {(end - t0) / 1e3:8.1f} "
          f"{(int(rows[b]['Start_Timestamp']) - end) / 1e3:8.1f}   {' '.join(tail)}")
