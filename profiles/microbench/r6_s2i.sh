mkdir -p gpurun_out/s2i
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "embed_grad" > gpurun_out/s2i/test_k.log 2>&1; tail -12 gpurun_out/s2i/test_k.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/s2i/$name.json 2>gpurun_out/s2i/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s2i/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run cur X=1
run olddups RP_SMP_DUPS=0
run cur2 X=1
B="$B --id-dist zipf"; run zipf X=1; run zipf_olddups RP_SMP_DUPS=0
