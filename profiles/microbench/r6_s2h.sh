mkdir -p gpurun_out/s2h
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/s2h/$name.json 2>gpurun_out/s2h/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s2h/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run cur X=1
run ahead RP_CATCHUP_AHEAD=1
run cur2 X=1
run ahead2 RP_CATCHUP_AHEAD=1
