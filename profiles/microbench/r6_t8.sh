mkdir -p gpurun_out/t8
B="python bench.py --no-cpu-baseline --no-small-batch --long-steps 300"
run() { name=$1; shift; timeout 400 $B "$@" > gpurun_out/t8/$name.json 2>gpurun_out/t8/$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/t8/$name.json").read().strip().splitlines()[-1])
    lr=d.get("long_run") or {}
    print("$name", d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"))
except Exception as e: print("$name ERR", e)
PY
}
run deepfm
run zipf --id-dist zipf
run exact --replay exact
run b8192 --batch 8192
run deepfm2
