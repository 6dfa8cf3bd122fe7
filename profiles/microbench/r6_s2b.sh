mkdir -p gpurun_out/s2b
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "embed_grad" > gpurun_out/s2b/test_k.log 2>&1; tail -4 gpurun_out/s2b/test_k.log
python -m pytest tests/test_hip_graph.py tests/test_hip_models.py -x -q -m gpu > gpurun_out/s2b/test_m.log 2>&1; tail -4 gpurun_out/s2b/test_m.log
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/s2b/$name.json 2>gpurun_out/s2b/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s2b/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run cur X=1
run noahead RP_SS_MARK_AHEAD=0
run cur2 X=1
