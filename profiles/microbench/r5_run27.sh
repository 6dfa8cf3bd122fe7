ulimit -c 0
mkdir -p gpurun_out/r5aa
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_models.py tests/test_hip_blocks.py -q -m gpu -k "mmoe or omoe or sharebottom or moe" > gpurun_out/r5aa/pytest.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5aa/pytest.txt | head -20
timeout 300 python bench.py --model mmoe --no-cpu-baseline --no-small-batch --long-steps 300 2>gpurun_out/r5aa/b_mmoe.err | grep "^{" > gpurun_out/r5aa/b_mmoe.json
grep -i "fell back\|why\|Error" gpurun_out/r5aa/b_mmoe.err | head -5
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5aa/b_mmoe.json")); print("mmoe", d["ms_per_step"], d["config"]["captured_step_backend"], (d.get("long_run") or {}).get("mean_ms"), str(d["config"].get("hip_graph"))[-200:])
PY
timeout 300 python bench.py --model mmoe --graph off --no-cpu-baseline --no-small-batch --long-steps 0 2>/dev/null | grep "^{" > gpurun_out/r5aa/b_mmoe_eager.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5aa/b_mmoe_eager.json")); print("mmoe eager", d["ms_per_step"])
PY
