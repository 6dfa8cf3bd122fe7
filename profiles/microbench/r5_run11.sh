set -x
ulimit -c 0
mkdir -p gpurun_out/r5k gpurun_out/profiles
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r5k/t_all.log 2>&1
grep -n "passed\|failed" gpurun_out/r5k/t_all.log | tail -3
grep -n "^FAILED" gpurun_out/r5k/t_all.log | head -20
timeout 600 python bench.py > gpurun_out/r5k/bench_default.json 2> gpurun_out/r5k/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5k/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "long", d["long_run"]["mean_ms"], d["long_run"]["p99_ms"], "host", d["host_call_ms_per_step_unblocked"], d["long_run"]["host_call_ms_per_step_unblocked"])
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["duration_ms"], "phase", d["roofline_phase"]["phase_ms"])
print("parity", d["full_size_parity"]["ok"], "cpu", d["cpu_baseline"]["value"])
PY
