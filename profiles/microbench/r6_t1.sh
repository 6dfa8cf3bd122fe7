cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ulimit -c 0
bash profiles/collect.sh r06_deepfm 1000
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/profiles/r06_trace_step.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
tail -5 gpurun_out/profiles/r06_trace_step.txt
