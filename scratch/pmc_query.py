"""Scratch: print PMC counter sums per kernel from a rocprofv3 results.db.  usage: pmc_query.py <db> [name filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()][:10])
try:
    cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
    print(cols)
    q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
    for r in c.execute(q):
        if flt in r[0]:
            print(f"{r[0][:50]:50s} {r[1]:28s} {r[2]:16.1f} n={r[3]}")
except Exception as e:
    print("ERR", e)
