mkdir -p gpurun_out/r6l
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/r6l/$name.json 2>/dev/null; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r6l/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run base_nosmp RP_GRAD_SMP=0
run smp X=1
run smp_side2normal RP_SIDE2_PRIORITY=normal
run smp_seg64 RP_SEG_ROWS=64
run smp_min4 RP_SMP_MIN=4
run base_nosmp2 RP_GRAD_SMP=0
run smp2 X=1
