mkdir -p gpurun_out/t5
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300 --sharded"
run() { name=$1; shift; env "$@" timeout 600 $B > gpurun_out/t5/$name.json 2>gpurun_out/t5/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t5/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), d["config"].get("captured_step_backend"), d.get("host_call_ms_per_step_unblocked"))
except Exception as e: print("$name ERR", e)
PY
}
run low X=1
run high RP_ROUTE_STREAM=high
run plain RP_ROUTE_STREAM=plain
run high2 RP_ROUTE_STREAM=high
