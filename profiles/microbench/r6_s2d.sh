mkdir -p gpurun_out/s2d
python -m pytest tests -x -q -m gpu > gpurun_out/s2d/test.log 2>&1; tail -15 gpurun_out/s2d/test.log
