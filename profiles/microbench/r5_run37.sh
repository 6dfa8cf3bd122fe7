ulimit -c 0
mkdir -p gpurun_out/r5ak
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
RP_PLAN_FORK=start timeout 300 $B > gpurun_out/r5ak/b_forkstart.json 2>/dev/null
timeout 300 $B > gpurun_out/r5ak/b_base.json 2>/dev/null
RP_PLAN_FORK=start timeout 300 $B > gpurun_out/r5ak/b_forkstart2.json 2>/dev/null
timeout 300 $B > gpurun_out/r5ak/b_base2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5ak/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
    except Exception as e: print(f, "ERR", e)
PY
