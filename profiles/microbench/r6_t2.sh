mkdir -p gpurun_out/t2
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; timeout 400 $B "$@" > gpurun_out/t2/$name.json 2>gpurun_out/t2/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t2/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run pre8 --pre-window 8
run pre64 --pre-window 64
run pre400 --pre-window 400
run pre8b --pre-window 8
run pre64b --pre-window 64
