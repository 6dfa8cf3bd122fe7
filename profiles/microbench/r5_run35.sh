ulimit -c 0
export TMPDIR=/tmp
mkdir -p gpurun_out/r5ai gpurun_out/profiles
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r5ai/pytest_full.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5ai/pytest_full.txt | head -20
bash profiles/collect.sh r05_deepfm 1000
rm -rf gpurun_out/prof_trace; mkdir -p gpurun_out/prof_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -o t -- python profiles/microbench/probes/probe_plan_longrun.py > gpurun_out/prof_trace/log.txt 2>&1
python profiles/trace_step.py gpurun_out/prof_trace 1050 > gpurun_out/profiles/r05_trace_step.txt 2>&1
python profiles/trace_periods.py gpurun_out/prof_trace > gpurun_out/profiles/r05_trace_periods.txt 2>&1
find gpurun_out/prof_trace -name "*.csv" -size +1M -delete; find gpurun_out/prof_trace -name "*.db" -delete
cat gpurun_out/profiles/r05_trace_step.txt
