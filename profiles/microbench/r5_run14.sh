set -x
ulimit -c 0
mkdir -p gpurun_out/r5n
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_graph.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5n/b_light.json 2>gpurun_out/r5n/b_light.err
RP_PLAN_EVENT_FENCE=system RP_STEP_MARKERS=torch timeout 300 $B > gpurun_out/r5n/b_sysfence.json 2>/dev/null
RP_STEP_MARKERS=torch timeout 300 $B > gpurun_out/r5n/b_torchmark.json 2>/dev/null
timeout 300 $B > gpurun_out/r5n/b_light2.json 2>/dev/null
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/r5n/trace_step.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5n/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "hostmax", d.get("host_call_max_ms_in_window"), d.get("host_stall"), d.get("host_slowest_step_in_window"), d.get("gc_in_window"))
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r5n/trace_step.txt
