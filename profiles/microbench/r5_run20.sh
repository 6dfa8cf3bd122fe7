ulimit -c 0
mkdir -p gpurun_out/r5t
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "mlp_tail" > gpurun_out/r5t/pytest_tail.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5t/pytest_tail.txt | head -20
timeout 1200 python -m pytest tests/test_hip_graph.py tests/test_hip_models.py -x -q -m gpu > gpurun_out/r5t/pytest_models.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5t/pytest_models.txt | head -20
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5t/b_fused.json 2>gpurun_out/r5t/b_fused.err
RP_TAIL_BCE=0 timeout 300 $B > gpurun_out/r5t/b_sep.json 2>/dev/null
RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5t/b_fused_ahead.json 2>/dev/null
RP_TAIL_BCE=0 RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5t/b_sep_ahead.json 2>/dev/null
timeout 300 $B > gpurun_out/r5t/b_fused2.json 2>/dev/null
tail -3 gpurun_out/r5t/b_fused.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5t/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["config"]["captured_step_backend"], (d.get("full_size_parity") or {}).get("ok"))
    except Exception as e: print(f, "ERR", e)
PY
