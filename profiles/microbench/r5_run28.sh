ulimit -c 0
mkdir -p gpurun_out/r5ab
timeout 900 python -m pytest tests/test_hip_graph.py -q -m gpu -k "mmoe" > gpurun_out/r5ab/pytest.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5ab/pytest.txt | head -20
