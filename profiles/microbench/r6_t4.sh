mkdir -p gpurun_out/t4
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "route_kernels" > gpurun_out/t4/test_k.log 2>&1; tail -15 gpurun_out/t4/test_k.log | cut -c1-300
python -m pytest tests/test_hip_graph.py -x -q -m gpu -k "sharded" > gpurun_out/t4/test_g.log 2>&1; tail -25 gpurun_out/t4/test_g.log | cut -c1-400
python -m pytest tests/test_hip_sharded_world2.py -x -q -m gpu > gpurun_out/t4/test_w2.log 2>&1; tail -25 gpurun_out/t4/test_w2.log | cut -c1-400
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300 --sharded"
run() { name=$1; shift; env "$@" timeout 600 $B > gpurun_out/t4/$name.json 2>gpurun_out/t4/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t4/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), d["config"].get("captured_step_backend"), d.get("host_call_ms_per_step_unblocked"))
except Exception as e: print("$name ERR", e)
PY
tail -3 gpurun_out/t4/$name.err | cut -c1-300
}
run sharded_plan X=1
run sharded_noahead RP_SHARD_AHEAD=0
run sharded_noseg RP_GRAD_SEG=0
