#!/bin/bash
# Round-5 bench lines of the secondary configurations (run on the GPU box via gpurun; output: gpurun_out/profiles/*.json).
# `wide_bf16`: the single-product bf16 mode of the matrix-core-bound MLP variant (MFMA utilisation at one
# product per flop next to the parity mode); `bf16_train`: the bf16-storage training line (row n2).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles
run() { tag=$1; shift; timeout 170 python bench.py --no-cpu-baseline --no-small-batch --long-steps 300 "$@" 2>/dev/null | grep "^{" > gpurun_out/profiles/r06_bench_$tag.json
  python - gpurun_out/profiles/r06_bench_$tag.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"] or {}; g=d.get("roofline_gemm") or {}
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], d["value"], d["config"].get("captured_step_backend"),
          r.get("kernel"), r.get("frac"), "| gemm", g.get("kernel"), g.get("achieved"), g.get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run eager --graph off
run immediate --defer off --graph off
run exact_replay --replay exact
run b8192 --batch 8192
run b8192_eager --batch 8192 --graph off
run zipf --id-dist zipf
run fwd --mode forward
run fwd_bf16 --mode forward --storage bf16
run bf16_train --storage bf16
run dcn --model dcn
run autoint --model autoint
run mmoe --model mmoe
run wide --hidden 1024,512,256
run wide_bf16 --hidden 1024,512,256 --precision bf16
run xdeepfm --model xdeepfm
run sharded --sharded
