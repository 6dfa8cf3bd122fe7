#!/usr/bin/env python
"""average duration per kernel out of a rocprofv3 --kernel-trace --stats run: python profiles/kstats.py <dir> [regex]"""
import csv
import os
import re
import sys

pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("kernel_stats.csv"):
            for r in csv.DictReader(open(os.path.join(root, f))):
                if pat.search(r["Name"]):
                    n = r["Name"].replace("void ", "")
                    n = n[:n.find("(")] if "(" in n else n
                    print(f"{n[:60]:60s} {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us  (min {int(r['MinNs']) / 1e3:.1f}, max {int(r['MaxNs']) / 1e3:.1f})  {float(r['Percentage']):5.2f} %")
