ulimit -c 0
mkdir -p gpurun_out/r5v
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
RP_SEG_FIRST=1 timeout 300 $B > gpurun_out/r5v/b_segfirst.json 2>/dev/null
timeout 300 $B > gpurun_out/r5v/b_base.json 2>/dev/null
RP_SEG_FIRST=1 timeout 300 $B > gpurun_out/r5v/b_segfirst2.json 2>/dev/null
timeout 300 $B > gpurun_out/r5v/b_base2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5v/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["config"]["captured_step_backend"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5v/pytest_full.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5v/pytest_full.txt | head -20
