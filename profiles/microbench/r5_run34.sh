ulimit -c 0
mkdir -p gpurun_out/r5ah
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
RP_PLAN_REBIND=0 timeout 300 $B > gpurun_out/r5ah/b_copy.json 2>/dev/null
timeout 300 $B > gpurun_out/r5ah/b_rebind.json 2>/dev/null
RP_PLAN_REBIND=0 timeout 300 $B > gpurun_out/r5ah/b_copy2.json 2>/dev/null
timeout 300 $B > gpurun_out/r5ah/b_rebind2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5ah/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        big=[(r["kernel"][:20], r["ms"]) for r in (d.get("in_step_launches") or []) if r["ms"]>0.1 and r["stream"]=="main"]
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "host", d["host_call_ms_per_step_unblocked"], big)
    except Exception as e: print(f, "ERR", e)
PY
