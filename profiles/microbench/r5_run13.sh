set -x
ulimit -c 0
mkdir -p gpurun_out/r5m
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5m/b_base.json 2>/dev/null
RP_PLAN_JOIN_SORT=early timeout 300 $B > gpurun_out/r5m/b_joinearly.json 2>/dev/null
RP_SEG_OCC1=1 timeout 300 $B > gpurun_out/r5m/b_occ1.json 2>/dev/null
timeout 300 $B > gpurun_out/r5m/b_base2.json 2>/dev/null
for v in base occ1; do
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
if [ $v = occ1 ]; then export RP_SEG_OCC1=1; fi
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/r5m/trace_step_$v.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
done
unset RP_SEG_OCC1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5m/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        seg=[(r["kernel"][:22], r["ms"]) for r in (d.get("in_step_launches") or []) if r["ms"]>0.12]
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "hostmax", d.get("host_call_max_ms_in_window"), seg)
    except Exception as e: print(f, "ERR", e)
PY
for v in base occ1; do echo == $v; grep -n "mlp_tail_bwd_kernel\|embed_grad_seg_kernel\|adam_kernel\|steps in" gpurun_out/r5m/trace_step_$v.txt; done
