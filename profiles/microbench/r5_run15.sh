set -x
ulimit -c 0
mkdir -p gpurun_out/r5o
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5o/b_base.json 2>/dev/null
RP_SIDE2_PRIORITY=normal timeout 300 $B > gpurun_out/r5o/b_s2normal.json 2>/dev/null
RP_SIDE2_PRIORITY=high timeout 300 $B > gpurun_out/r5o/b_s2high.json 2>/dev/null
timeout 300 $B --steps 40 > gpurun_out/r5o/b_k40.json 2>/dev/null
for v in normal high; do
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
RP_SIDE2_PRIORITY=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/r5o/trace_step_$v.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5o/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "hostmax", d.get("host_call_max_ms_in_window"), d.get("host_stall"), d.get("host_slowest_step_in_window"))
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r5o/trace_step_normal.txt gpurun_out/r5o/trace_step_high.txt
