set -x
ulimit -c 0
mkdir -p gpurun_out/r5i
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r5i/t_all.log 2>&1
grep -n "passed\|failed" gpurun_out/r5i/t_all.log | tail -3
grep -n "^FAILED" gpurun_out/r5i/t_all.log | head -20
timeout 600 python profiles/microbench/probes/aten_sources.py mmoe autoint xdeepfm deepfm > gpurun_out/r5i/aten_sources.log 2>&1
grep -v amdgpu gpurun_out/r5i/aten_sources.log | head -150
