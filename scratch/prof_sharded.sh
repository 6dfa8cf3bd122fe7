#!/bin/bash
# rocprofv3 kernel stats of the sharded path at G = 1 (every kernel, torch's own included)
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/prof_sh
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sh -o s -- python bench.py --sharded --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_sh.log 2>&1
F=$(find gpurun_out/prof_sh -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot/1e6)
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
