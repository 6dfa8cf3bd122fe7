ulimit -c 0
mkdir -p gpurun_out/r5ag
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_graph.py -q -m gpu > gpurun_out/r5ag/pytest_graph.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5ag/pytest_graph.txt | head -20
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5ag/b_rebind.json 2>gpurun_out/r5ag/b_rebind.err
tail -3 gpurun_out/r5ag/b_rebind.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5ag/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["config"]["captured_step_backend"], "host", d["host_call_ms_per_step_unblocked"])
    except Exception as e: print(f, "ERR", e)
PY
