#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles
for m in xdeepfm autoint; do
  python bench.py --model $m --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/profiles/r01_bench_n1_$m.json
  python - "gpurun_out/profiles/r01_bench_n1_$m.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d["ms_per_step"], d["value"], d["roofline"])
PY
done
