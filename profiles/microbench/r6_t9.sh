mkdir -p gpurun_out/t9
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300 --window-events"
run() { name=$1; shift; timeout 400 $B "$@" > gpurun_out/t9/$name.json 2>gpurun_out/t9/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t9/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
grep "window events" gpurun_out/t9/$name.err | cut -c1-700
}
run zipf --id-dist zipf --steps 60
run exact --replay exact --steps 40
