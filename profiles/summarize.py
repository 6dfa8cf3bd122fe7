#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/prof_*/) into the small, tracked files under profiles/.

    python profiles/summarize.py <round tag, e.g. r01> <stats dir> [<fetch dir> <write dir>]

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table, our kernels + top-10 others) and
profiles/<tag>_pmc.json: per kernel mean FETCH_SIZE / WRITE_SIZE per launch in bytes.
Units and correction follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB
(bytes = value * 1024) and on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced
streaming read, so `fetch_bytes_corrected` = 2 x raw; WRITE_SIZE is used as reported.
"""
import collections
import csv
import json
import os
import sys

OURS = ("adam_kernel", "embed_", "linear_", "wgrad_", "transpose_kernel", "relu_bwd", "sigmoid_bce", "loss_finish",
        "zero_rows", "iota_i32", "cin_", "cross_", "attn_", "mmoe_", "radix_sort", "lazy_", "a2a_", "fm_", "bn_",
        "trampoline_kernel", "onesweep")


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0][:80]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    rows = list(csv.DictReader(open(os.path.join(stats_dir, [f for f in os.listdir(stats_dir)
                                                             if f.endswith("kernel_stats.csv")][0]))))
    keep = [r for r in rows if any(o in r["Name"] for o in OURS)]
    others = [r for r in rows if r not in keep][:10]
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in keep + others:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    if len(sys.argv) >= 5:
        pmc = collections.defaultdict(dict)
        for d, counter in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            fn = [f for f in os.listdir(d) if f.endswith("counter_collection.csv")][0]
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(os.path.join(d, fn))):
                if r["Counter_Name"] == counter and any(o in r["Kernel_Name"] for o in OURS):
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                b = sum(v) / len(v) * 1024.0
                if counter == "FETCH_SIZE":
                    pmc[k]["fetch_bytes_raw"] = round(b)
                    pmc[k]["fetch_bytes_corrected"] = round(2 * b)
                else:
                    pmc[k]["write_bytes"] = round(b)
                pmc[k]["launches_" + counter] = len(v)
        for k, v in pmc.items():
            if "fetch_bytes_corrected" in v and "write_bytes" in v:
                v["hbm_bytes_per_launch"] = v["fetch_bytes_corrected"] + v["write_bytes"]
        with open(os.path.join(here, f"{tag}_pmc.json"), "w") as f:
            json.dump(pmc, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
