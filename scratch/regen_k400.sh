#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles
python bench.py --steps 400 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/profiles/r01_bench_n1_deepfm_k400.json
python -c "
import json; d=json.load(open('gpurun_out/profiles/r01_bench_n1_deepfm_k400.json')); print(d['steps'], d['ms_per_step'], d['value'])"
