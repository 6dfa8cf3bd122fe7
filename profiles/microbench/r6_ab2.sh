mkdir -p gpurun_out/r6n
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/r6n/$name.json 2>/dev/null; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6n/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run base_nosmp RP_GRAD_SMP=0
run tinymain X=1
run tinymain_s2normal RP_SIDE2_PRIORITY=normal
run tinyside_s2normal RP_SIDE2_PRIORITY=normal RP_TINY_MAIN=0
run tinymain_s2high RP_SIDE2_PRIORITY=high
run base_nosmp2 RP_GRAD_SMP=0
run tinymain2 X=1
