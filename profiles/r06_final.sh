# the round's last check on the GPU box: the whole GPU suite, the smoke entry, then every profile of the round (profiles/r06_all.sh)
mkdir -p gpurun_out/tfull
python -m pytest tests -q -m gpu > gpurun_out/tfull/test.log 2>&1; grep -n "passed\|failed" gpurun_out/tfull/test.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
[ "$1" = "tests" ] || bash profiles/r06_all.sh
