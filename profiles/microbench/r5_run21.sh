ulimit -c 0
mkdir -p gpurun_out/r5u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_graph.py -x -q -m gpu > gpurun_out/r5u/pytest_graph.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5u/pytest_graph.txt | head -20
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5u/b_side.json 2>gpurun_out/r5u/b_side.err
RP_STAGE_SIDE=0 timeout 300 $B > gpurun_out/r5u/b_main.json 2>/dev/null
timeout 300 $B > gpurun_out/r5u/b_side2.json 2>/dev/null
RP_STAGE_SIDE=0 timeout 300 $B > gpurun_out/r5u/b_main2.json 2>/dev/null
RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5u/b_side_ahead.json 2>/dev/null; RP_STAGE_SIDE=0 RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5u/b_main_ahead.json 2>/dev/null
tail -3 gpurun_out/r5u/b_side.err
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/r5u/trace_step.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5u/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["config"]["captured_step_backend"])
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r5u/trace_step.txt
