"""Summarise a rocprofv3 results.db: per kernel name + grid -> calls, mean us.  usage: prof_query.py <db> [filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = ("select name, grid_x, grid_y, grid_z, count(*), avg(end-start), min(end-start) from kernels "
     "group by name, grid_x, grid_y, grid_z order by sum(end-start) desc")
for r in c.execute(q):
    n = r[0].replace("void ", "")
    if flt and flt not in n:
        continue
    print(f"{n[:60]:60s} grid=({r[1]},{r[2]},{r[3]}) calls={r[4]} mean={r[5]/1e3:.1f}us min={r[6]/1e3:.1f}us")
