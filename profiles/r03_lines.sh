#!/bin/bash
# Round-3 bench lines of the secondary configurations (run on the GPU box via gpurun; output: gpurun_out/profiles/*.json)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles
run() { tag=$1; shift; timeout 170 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep "^{" > gpurun_out/profiles/r03_bench_$tag.json
  python - gpurun_out/profiles/r03_bench_$tag.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"] or {}
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["value"], r.get("kernel"), r.get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run exact_replay --replay exact
run b8192_graph --batch 8192
run b8192_eager --batch 8192 --graph off
run zipf --id-dist zipf
run fwd --mode forward
run dcn --model dcn
run autoint --model autoint
run mmoe --model mmoe
run wide --hidden 1024,512,256
run xdeepfm --model xdeepfm
run sharded --sharded
