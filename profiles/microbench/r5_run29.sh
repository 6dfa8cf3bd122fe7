ulimit -c 0
mkdir -p gpurun_out/r5ac
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_models.py tests/test_hip_blocks.py -q -m gpu -k "autoint or attention" > gpurun_out/r5ac/pytest.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5ac/pytest.txt | head -20
timeout 300 python profiles/microbench/probes/aten_sources.py autoint > gpurun_out/r5ac/aten.log 2>&1
cut -c1-200 gpurun_out/r5ac/aten.log | tail -12
timeout 300 python bench.py --model autoint --graph on --no-cpu-baseline --no-small-batch --long-steps 0 2>gpurun_out/r5ac/b.err | grep "^{" > gpurun_out/r5ac/b_autoint.json
grep -i "fell back\|why\|Error" gpurun_out/r5ac/b.err | head -5
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5ac/b_autoint.json")); print("autoint", d["ms_per_step"], d["config"]["captured_step_backend"], str(d["config"].get("hip_graph"))[-220:])
PY
