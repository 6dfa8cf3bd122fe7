mkdir -p gpurun_out/t12
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 600"
run() { name=$1; shift; timeout 400 $B "$@" > gpurun_out/t12/$name.json 2>gpurun_out/t12/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t12/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), d["host_slowest_step_in_window"], d["host_stall"]["slowest_hip_call_in_replay"])
except Exception as e: print("$name ERR", e)
PY
}
run a; run b; run c; run d; run e
