#!/bin/bash
# Alternating A/B runs of bench.py on ONE box (boxes of the pool differ by 2-8 %: only runs of one gpurun call compare).
#   gpurun -- 'bash profiles/ab.sh <tag> "name|ENV=1 ENV2=x|--bench --flags" "name2||--id-dist zipf" ...'
# Every variant: window ms/step, mean / p99 over the 600 steps behind it, slowest host call of the window.
# Results: gpurun_out/<tag>/<name>.json (the full bench line) and one summary line per variant on stdout.
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for spec in "$@"; do
  IFS='|' read -r name envs flags <<< "$spec"
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-small-batch --long-steps 600 $flags \
      > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  python - "$name" gpurun_out/$TAG/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    lr = d.get("long_run") or {}
    print(sys.argv[1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "host max", d.get("host_call_max_ms_in_window"),
          d["config"].get("captured_step_backend"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
