mkdir -p gpurun_out/t10
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env $ENVV timeout 400 $B "$@" > gpurun_out/t10/$name.json 2>gpurun_out/t10/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t10/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
ENVV="X=1" run b8192_r5form --batch 8192
ENVV="RP_SMP_MIN_BATCH=1" run b8192_r6form --batch 8192
ENVV="X=1" run b16384_r5form --batch 16384
ENVV="RP_SMP_MIN_BATCH=1" run b16384_r6form --batch 16384
ENVV="RP_SMP_MIN_BATCH=100000" run b32768_r5form --batch 32768
ENVV="X=1" run b32768_r6form --batch 32768
ENVV="RP_GRAD_SMP=0" run zipf_r5form --id-dist zipf
ENVV="X=1" run zipf_r6form --id-dist zipf
