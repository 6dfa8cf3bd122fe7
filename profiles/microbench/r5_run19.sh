ulimit -c 0
mkdir -p gpurun_out/r5s
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_graph.py -x -q -m gpu -k "catch_up_ahead" > gpurun_out/r5s/pytest_ahead.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5s/pytest_ahead.txt | head -20
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5s/b_ahead.json 2>/dev/null
timeout 300 $B > gpurun_out/r5s/b_base.json 2>/dev/null
RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5s/b_ahead2.json 2>/dev/null
timeout 300 $B > gpurun_out/r5s/b_base2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5s/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["config"]["captured_step_backend"])
    except Exception as e: print(f, "ERR", e)
PY
