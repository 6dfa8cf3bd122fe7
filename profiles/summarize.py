#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/prof_*/) into the small, tracked files under profiles/.

    python profiles/summarize.py <tag, e.g. r02_deepfm> <stats dir> [<fetch dir> <write dir>]

Writes
  profiles/<tag>_kernel_stats.csv   the --kernel-trace --stats table (our kernels + the top-10 others), plus — when the
                                    kernel trace is there — one `steady` row set: mean duration of the LAST 100 launches
                                    of every (kernel, grid size), i.e. the long-run state after bench.py's pre-roll;
  profiles/<tag>_pmc.json           per (kernel name | grid size): mean FETCH_SIZE / WRITE_SIZE per launch in bytes over
                                    the last 100 launches, and `mean_ns` from the kernel trace (bench.py matches its
                                    per-shape rows to these entries by kernel name + duration).
Units and correction follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB (bytes = value *
1024) and on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced streaming read, so
`fetch_bytes_corrected` = 2 x raw; WRITE_SIZE is used as reported.  Separate --pmc passes, no tracing domains mixed in.
"""
import collections
import csv
import json
import os
import sys

OURS = ("adam_kernel", "embed_", "linear_", "wgrad_", "transpose_kernel", "relu_bwd", "sigmoid_bce", "loss_finish",
        "zero_rows", "iota_i32", "cin_", "crossnet", "attn_", "mmoe_", "radix_sort", "lazy_", "a2a_", "fm_", "bn_",
        "trampoline_kernel", "onesweep", "field_sort", "sort_hist", "sort_scan", "sort_scatter", "mlp_tail", "dropout", "route_", "shard_", "batchnorm", "counter_add",
        "accumulate", "pool_", "multi_copy", "copy_rows", "dice_")
LAST = 100


def short(name):
    """kernel name without its parameter list, template arguments kept (they may contain parentheses: `float
    __vector(4)`), so that instantiations stay apart"""
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:110]


def find(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def grid_of(r):
    if r.get("Grid_Size"):
        return int(r["Grid_Size"])
    g = 1
    for ax in "XYZ":
        g *= int(r.get(f"Grid_Size_{ax}", 1) or 1)
    return g


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    rows = list(csv.DictReader(open(find(stats_dir, "kernel_stats.csv"))))
    keep = [r for r in rows if any(o in r["Name"] for o in OURS)]
    others = [r for r in rows if r not in keep][:10]
    steady = {}
    trace = find(stats_dir, "kernel_trace.csv")
    if trace:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            if any(o in r["Kernel_Name"] for o in OURS):
                name = short(r["Kernel_Name"])
                # (the closed-form table kernel's grid grows with the step count: one row, not one per grid size)
                key = name if name.startswith("lazy_cf_table_kernel") else f"{name}|g{grid_of(r)}"
                acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        steady = {k: (len(v), sum(v[-LAST:]) / len(v[-LAST:])) for k, v in acc.items()}
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in keep + others:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
        if steady:
            w.writerow([])
            w.writerow([f"# long-run state: mean of the last {LAST} launches per (kernel | grid size)"])
            w.writerow(["Name|grid", "Calls", "", "AverageNs_last%d" % LAST])
            for k, (n, ns) in sorted(steady.items(), key=lambda kv: -kv[1][1] * min(kv[1][0], LAST)):
                w.writerow([k, n, "", round(ns, 1)])
    if len(sys.argv) >= 5:
        pmc = collections.defaultdict(dict)
        for d, counter in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
                if r["Counter_Name"] == counter and any(o in r["Kernel_Name"] for o in OURS):
                    acc[f"{short(r['Kernel_Name'])}|g{grid_of(r)}"].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                tail = v[-LAST:]
                b = sum(tail) / len(tail) * 1024.0
                if counter == "FETCH_SIZE":
                    pmc[k]["fetch_bytes_raw"] = round(b)
                    pmc[k]["fetch_bytes_corrected"] = round(2 * b)
                else:
                    pmc[k]["write_bytes"] = round(b)
                pmc[k]["launches_" + counter] = len(v)
        for k, v in pmc.items():
            if "fetch_bytes_corrected" in v and "write_bytes" in v:
                v["hbm_bytes_per_launch"] = v["fetch_bytes_corrected"] + v["write_bytes"]
            if k in steady:
                v["mean_ns"] = round(steady[k][1], 1)
        with open(os.path.join(here, f"{tag}_pmc.json"), "w") as f:
            json.dump(pmc, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
