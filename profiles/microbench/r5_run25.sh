ulimit -c 0
mkdir -p gpurun_out/r5y
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_blocks.py tests/test_hip_models.py tests/test_hip_graph.py -q -m gpu > gpurun_out/r5y/pytest.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5y/pytest.txt | head -20
