mkdir -p gpurun_out/t11
( time python bench.py > gpurun_out/t11/default.json 2> gpurun_out/t11/default.err ) 2> gpurun_out/t11/time.txt
tail -3 gpurun_out/t11/time.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/t11/default.json").read().strip().splitlines()[-1])
lr=d.get("long_run") or {}
print(d["ms_per_step"], d["value"], lr.get("mean_ms"), lr.get("p99_ms"), d.get("pre_window_replays"))
print(json.dumps(d["roofline"])[:500])
print(json.dumps(d["cpu_baseline"])[:400])
print(json.dumps(d.get("lazy_adam"))[:600])
print(json.dumps(d.get("full_size_parity"))[:300])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
