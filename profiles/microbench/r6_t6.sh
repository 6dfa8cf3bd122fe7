mkdir -p gpurun_out/t6
timeout 300 python profiles/microbench/probes/aten_sources.py xdeepfm > gpurun_out/t6/aten_xdeepfm.txt 2>&1
grep -v "^W10\|amdgpu.ids" gpurun_out/t6/aten_xdeepfm.txt | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-small-batch --long-steps 100 --model xdeepfm 2>/dev/null | grep "^{" > gpurun_out/t6/xdeepfm.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/t6/xdeepfm.json")); print("xdeepfm", d["ms_per_step"], d["config"].get("captured_step_backend"))
for k,v in sorted((d.get("kernels") or {}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:25]:
    print("  ", k, v["calls_per_step"], v["mean_ms"], v["ms_per_step"])
PY
