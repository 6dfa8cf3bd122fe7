set -x
mkdir -p gpurun_out/r5d
export TMPDIR=/tmp
timeout 600 python profiles/microbench/probes/diag_fullsize.py > gpurun_out/r5d/diag.log 2>&1
timeout 900 python -m pytest tests/test_hip_deferred_adam.py tests/test_hip_optim_vs_oracle.py tests/test_hip_lazy_adam.py -x -q > gpurun_out/r5d/t_adam.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --long-steps 500 > gpurun_out/r5d/bench_wave.json 2> gpurun_out/r5d/bench_wave.err
RP_CATCHUP_WAVE=0 timeout 600 python bench.py --no-cpu-baseline --long-steps 500 > gpurun_out/r5d/bench_nowave.json 2> gpurun_out/r5d/bench_nowave.err
cat gpurun_out/r5d/diag.log | grep -v amdgpu.ids
tail -n 3 gpurun_out/r5d/t_adam.log
python - <<'PY'
import json
for n in ("wave","nowave"):
    d=json.loads(open(f"gpurun_out/r5d/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["long_run"]["mean_ms"], [ (r["kernel"][:28], r["ms"]) for r in d["in_step_launches"] if r["ms"]>0.03 and r["stream"]!="side"])
PY
