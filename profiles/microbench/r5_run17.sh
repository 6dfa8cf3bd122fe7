set -x
ulimit -c 0
mkdir -p gpurun_out/r5q
export TMPDIR=/tmp
RP_CATCHUP_AHEAD=1 timeout 900 python -m pytest tests/test_hip_graph.py -x -q -m gpu > gpurun_out/r5q/pytest_graph_ahead.txt 2>&1
tail -5 gpurun_out/r5q/pytest_graph_ahead.txt
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5q/b_ahead.json 2>gpurun_out/r5q/b_ahead.err
timeout 300 $B > gpurun_out/r5q/b_base.json 2>/dev/null
RP_CATCHUP_AHEAD=1 timeout 300 $B > gpurun_out/r5q/b_ahead2.json 2>/dev/null
timeout 300 $B > gpurun_out/r5q/b_base2.json 2>/dev/null
tail -3 gpurun_out/r5q/b_ahead.err
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
RP_CATCHUP_AHEAD=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1; tail -3 gpurun_out/prof_trace/log.txt
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/r5q/trace_step_ahead.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5q/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "hostmax", d.get("host_call_max_ms_in_window"), d.get("full_size_parity",{}).get("ok") if isinstance(d.get("full_size_parity"),dict) else d.get("full_size_parity"))
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r5q/trace_step_ahead.txt
