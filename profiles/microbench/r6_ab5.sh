mkdir -p gpurun_out/r6w
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/r6w/$name.json 2>/dev/null; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6w/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run base_nosmp RP_GRAD_SMP=0
run behind_main X=1
run behind_side RP_SMP_BEHIND=side
run behind_main2 X=1
run behind_side2 RP_SMP_BEHIND=side
