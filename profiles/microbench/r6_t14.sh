mkdir -p gpurun_out/t14
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "sort" > gpurun_out/t14/test_k.log 2>&1; tail -5 gpurun_out/t14/test_k.log | cut -c1-300
python -m pytest tests/test_hip_graph.py tests/test_hip_models.py -x -q -m gpu > gpurun_out/t14/test_m.log 2>&1; tail -5 gpurun_out/t14/test_m.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 600"
run() { name=$1; shift; env $ENVV timeout 400 $B "$@" > gpurun_out/t14/$name.json 2>gpurun_out/t14/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t14/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    k=(d.get("kernels") or {}).get("sort_pairs_i32") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "sort alone ms", k.get("mean_ms"))
except Exception as e: print("$name ERR", e)
PY
}
ENVV="X=1" run new1
ENVV="RP_SORT_FIELDS=0" run old1
ENVV="X=1" run new2
ENVV="RP_SORT_FIELDS=0" run old2
