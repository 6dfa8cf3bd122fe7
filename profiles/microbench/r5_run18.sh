ulimit -c 0
mkdir -p gpurun_out/r5r
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
export RP_CATCHUP_AHEAD=1
timeout 300 $B > gpurun_out/r5r/b_a_start_low.json 2>/dev/null
RP_PLAN_FORK=backward timeout 300 $B > gpurun_out/r5r/b_a_bwd_low.json 2>/dev/null
RP_PLAN_FORK=backward RP_SIDE_PRIORITY=normal RP_SIDE2_PRIORITY=low timeout 300 $B > gpurun_out/r5r/b_a_bwd_normal.json 2>/dev/null
RP_SIDE_PRIORITY=normal RP_SIDE2_PRIORITY=low timeout 300 $B > gpurun_out/r5r/b_a_start_normal.json 2>/dev/null
unset RP_CATCHUP_AHEAD
timeout 300 $B > gpurun_out/r5r/b_base.json 2>/dev/null
RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5r/b_a_start_low2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5r/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["config"]["captured_step_backend"])
    except Exception as e: print(f, "ERR", e)
PY
