set -x
ulimit -c 0
mkdir -p gpurun_out/r5h
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5h/b_segfirst.json 2>gpurun_out/r5h/b_segfirst.err
RP_SEG_FIRST=0 timeout 300 $B > gpurun_out/r5h/b_segsecond.json 2>/dev/null
timeout 300 $B > gpurun_out/r5h/b_segfirst2.json 2>/dev/null
RP_SEG_FIRST=0 timeout 300 $B > gpurun_out/r5h/b_segsecond2.json 2>/dev/null
timeout 300 $B --model dcn > gpurun_out/r5h/b_dcn.json 2>gpurun_out/r5h/b_dcn.err
timeout 300 $B --storage bf16 > gpurun_out/r5h/b_bf16.json 2>gpurun_out/r5h/b_bf16.err
timeout 300 python bench.py --no-cpu-baseline --long-steps 0 --no-small-batch --mode forward --storage bf16 > gpurun_out/r5h/b_fwd_bf16.json 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5h/t_all.log 2>&1
grep -n "passed\|failed" gpurun_out/r5h/t_all.log | tail -3
tail -3 gpurun_out/r5h/b_dcn.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5h/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        seg=[(r["kernel"][:22], r["ms"]) for r in (d.get("in_step_launches") or []) if r["ms"]>0.09]
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["config"].get("captured_step_backend"), (d.get("roofline_phase") or {}).get("phase_ms"), seg, (d.get("roofline_gather") or {}).get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
