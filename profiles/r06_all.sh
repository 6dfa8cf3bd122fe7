#!/bin/bash
# Everything round 5 measures on the GPU box besides the test suite (run via gpurun; results come back through gpurun_out/profiles/).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ulimit -c 0
bash profiles/collect.sh r06_deepfm 1000
# timeline of replayed steps (kernel trace of the long-run probe)
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/profiles/r06_trace_step.txt 2>&1
python profiles/trace_periods.py gpurun_out/prof_trace > gpurun_out/profiles/r06_trace_periods.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
tail -5 gpurun_out/profiles/r06_trace_step.txt
bash profiles/mfma_util.sh r06_deepfm_wide --hidden 1024,512,256
bash profiles/mfma_util.sh r06_mmoe --model mmoe
bash profiles/mfma_util.sh r06_xdeepfm --model xdeepfm
bash profiles/r06_lines.sh
