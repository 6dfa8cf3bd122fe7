set -x
mkdir -p gpurun_out/r5e
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5e/t_all.log 2>&1
timeout 300 python profiles/microbench/probes/aten_in_step.py > gpurun_out/r5e/aten.log 2>&1
timeout 600 python bench.py > gpurun_out/r5e/bench.json 2> gpurun_out/r5e/bench.err
tail -n 5 gpurun_out/r5e/t_all.log
grep -v amdgpu.ids gpurun_out/r5e/aten.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5e/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["long_run"]["mean_ms"])
print("parity", d.get("full_size_parity"))
print("cpu", d.get("cpu_baseline"))
PY
