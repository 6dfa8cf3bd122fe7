ulimit -c 0
mkdir -p gpurun_out/r5x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_blocks.py tests/test_hip_models.py tests/test_hip_graph.py -x -q -m gpu > gpurun_out/r5x/pytest.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5x/pytest.txt | head -20
timeout 300 python profiles/microbench/probes/aten_sources.py mmoe > gpurun_out/r5x/aten_mmoe.log 2>&1
cut -c1-200 gpurun_out/r5x/aten_mmoe.log | tail -25
timeout 300 python bench.py --model mmoe --no-cpu-baseline --no-small-batch --long-steps 0 2>/dev/null | grep "^{" > gpurun_out/r5x/b_mmoe.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5x/b_mmoe.json")); print("mmoe", d["ms_per_step"], d["config"]["captured_step_backend"])
ks=d["kernels"]
for n,k in sorted(ks.items(), key=lambda kv:-kv[1]["ms_per_step"])[:12]: print(f'{n:48s} {k["calls_per_step"]:4.1f} {k["mean_ms"]:.4f} {k["ms_per_step"]:.4f}')
PY
