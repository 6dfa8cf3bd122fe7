mkdir -p gpurun_out/r6u
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/r6u/$name.json 2>/dev/null; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6u/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run base_nosmp RP_GRAD_SMP=0
run cur X=1
run early RP_TINY_EARLY=1
run early_s2normal RP_TINY_EARLY=1 RP_SIDE2_PRIORITY=normal
run early_s2high RP_TINY_EARLY=1 RP_SIDE2_PRIORITY=high
run cur2 X=1
run early2 RP_TINY_EARLY=1
