"""per-kernel statistics (calls, mean / min / max us) from a rocprofv3 rocpd .db (the default output of ROCm 7.2's
rocprofv3 --kernel-trace): python profiles/dbstats.py <file.db> [name substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for n, s, e in rows:
    n = n.split("(")[0]
    a = agg.setdefault(n, [])
    a.append((e - s) / 1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':70s} {'calls':>6s} {'mean us':>9s} {'min':>8s} {'max':>8s} {'share':>6s}")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if flt in n:
        print(f"{n[:70]:70s} {len(v):6d} {sum(v) / len(v):9.1f} {min(v):8.1f} {max(v):8.1f} {100 * sum(v) / tot:5.1f}%")
