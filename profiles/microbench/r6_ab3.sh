mkdir -p gpurun_out/r6t
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/r6t/$name.json 2>/dev/null; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6t/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run base_nosmp RP_GRAD_SMP=0
run cur X=1
run s2normal RP_SIDE2_PRIORITY=normal
run s2normal_tinyside RP_SIDE2_PRIORITY=normal RP_TINY_MAIN=0
run tinyside RP_TINY_MAIN=0
run s2high RP_SIDE2_PRIORITY=high
run base_nosmp2 RP_GRAD_SMP=0
run cur2 X=1
