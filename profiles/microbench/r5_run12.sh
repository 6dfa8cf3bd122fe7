set -x
ulimit -c 0
mkdir -p gpurun_out/r5l
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5l/b_base.json 2>/dev/null
RP_PLAN_FORK=start timeout 300 $B > gpurun_out/r5l/b_forkstart.json 2>/dev/null
timeout 300 $B > gpurun_out/r5l/b_base2.json 2>/dev/null
RP_PLAN_FORK=start timeout 300 $B > gpurun_out/r5l/b_forkstart2.json 2>/dev/null
RP_SEG_TILES=4 timeout 300 $B > gpurun_out/r5l/b_T4.json 2>/dev/null
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/r5l/trace_step.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5l/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        seg=[(r["kernel"][:22], r["ms"]) for r in (d.get("in_step_launches") or []) if r["ms"]>0.12]
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "hostmax", d.get("host_call_max_ms_in_window"), d.get("host_call_ms_per_step_unblocked"), seg)
    except Exception as e: print(f, "ERR", e)
PY
head -40 gpurun_out/r5l/trace_step.txt
