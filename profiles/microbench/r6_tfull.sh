mkdir -p gpurun_out/tfull
python -m pytest tests -q -m gpu > gpurun_out/tfull/test.log 2>&1; tail -15 gpurun_out/tfull/test.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-small-batch --long-steps 300 --sharded 2>/dev/null | grep "^{" > gpurun_out/tfull/sharded.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/tfull/sharded.json")); print("sharded", d["ms_per_step"], (d.get("long_run") or {}).get("mean_ms"))
PY
timeout 300 python bench.py --no-cpu-baseline --no-small-batch --long-steps 300 2>/dev/null | grep "^{" > gpurun_out/tfull/deepfm.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/tfull/deepfm.json")); print("deepfm", d["ms_per_step"], (d.get("long_run") or {}).get("mean_ms"), d.get("pre_window_replays"))
PY
