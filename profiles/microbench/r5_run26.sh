ulimit -c 0
mkdir -p gpurun_out/r5z
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_models.py tests/test_hip_blocks.py -q -m gpu -k "mmoe or omoe or sharebottom or moe" > gpurun_out/r5z/pytest.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5z/pytest.txt | head -20
timeout 300 python profiles/microbench/probes/aten_sources.py mmoe > gpurun_out/r5z/aten_mmoe.log 2>&1
cut -c1-200 gpurun_out/r5z/aten_mmoe.log | tail -12
timeout 300 python bench.py --model mmoe --no-cpu-baseline --no-small-batch --long-steps 0 2>gpurun_out/r5z/b_mmoe.err | grep "^{" > gpurun_out/r5z/b_mmoe.json
grep -i "fell back\|why" gpurun_out/r5z/b_mmoe.err | head -3
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5z/b_mmoe.json")); print("mmoe", d["ms_per_step"], d["config"]["captured_step_backend"], str(d["config"].get("hip_graph"))[-160:])
PY
