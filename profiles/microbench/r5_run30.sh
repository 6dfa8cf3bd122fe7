ulimit -c 0
mkdir -p gpurun_out/r5ad
timeout 900 python -m pytest tests/test_hip_graph.py -q -m gpu -k "autoint or mmoe" > gpurun_out/r5ad/pytest.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5ad/pytest.txt | head -20
