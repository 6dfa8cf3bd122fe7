set -x
mkdir -p gpurun_out/r5c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_kernels.py -k "embed_grad_seg or embed_grad_tiny" -x -q > gpurun_out/r5c/t_seg128.log 2>&1
timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5c/probe_128.log 2>&1
RP_SEG_TILES=4 timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5c/probe_128_T4.log 2>&1
timeout 900 python -m pytest tests/test_hip_models.py -x -q -k "full_size or bf16_storage" > gpurun_out/r5c/t_new.log 2>&1
timeout 900 python bench.py > gpurun_out/r5c/bench.json 2> gpurun_out/r5c/bench.err
for f in gpurun_out/r5c/t_*.log; do tail -n 3 $f; done
grep -H "round" gpurun_out/r5c/probe_*.log
tail -5 gpurun_out/r5c/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
print("roofline", json.dumps(d["roofline"])[:900])
print("phase", d["roofline_phase"])
print("long_run", d["long_run"])
print("host", d["host_enqueue_ms_per_step"], d["host_call_ms_per_step_unblocked"], d["host_wait_ms_per_step"])
print("parity", d.get("full_size_parity"))
print("cpu", d.get("cpu_baseline"))
for r in d["in_step_launches"]: print("   ", r)
PY
