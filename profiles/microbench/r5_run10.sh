set -x
ulimit -c 0
mkdir -p gpurun_out/r5j
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5j/b_base.json 2>/dev/null
RP_SIDE_PRIORITY=low timeout 300 $B > gpurun_out/r5j/b_low.json 2>gpurun_out/r5j/b_low.err
timeout 300 $B > gpurun_out/r5j/b_base2.json 2>/dev/null
RP_SIDE_PRIORITY=low timeout 300 $B > gpurun_out/r5j/b_low2.json 2>/dev/null
RP_SIDE_PRIORITY=low RP_SEG_FIRST=1 timeout 300 $B > gpurun_out/r5j/b_low_segfirst.json 2>/dev/null
timeout 900 python -m pytest tests/test_hip_models.py tests/test_hip_sharded_world2.py tests/test_hip_graph.py -q -k "sharded_fused or bf16_storage or two_ranks or dcn" > gpurun_out/r5j/t_fix.log 2>&1
grep -n "passed\|failed" gpurun_out/r5j/t_fix.log | tail -2
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5j/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        seg=[(r["kernel"][:22], r["ms"]) for r in (d.get("in_step_launches") or []) if r["ms"]>0.09]
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), (d.get("roofline_phase") or {}).get("phase_ms"), seg)
    except Exception as e: print(f, "ERR", e)
PY
