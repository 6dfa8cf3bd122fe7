mkdir -p gpurun_out/tfull
python -m pytest tests -q -m gpu > gpurun_out/tfull/test.log 2>&1; tail -6 gpurun_out/tfull/test.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
