set -x
ulimit -c 0
mkdir -p gpurun_out/r5p
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_lazy_adam.py tests/test_hip_graph.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5p/b_T8.json 2>/dev/null
for T in 6 7 9 10 11; do
RP_SEG_TILES=$T timeout 300 $B > gpurun_out/r5p/b_T$T.json 2>/dev/null
RP_SEG_TILES=$T timeout 120 python profiles/microbench/probes/probe_grad_seg.py 2>&1 | grep "round 5: embed_grad_seg" > gpurun_out/r5p/probe_T$T.log
done
timeout 120 python profiles/microbench/probes/probe_grad_seg.py 2>&1 | grep "round 5: embed_grad_seg" > gpurun_out/r5p/probe_T8.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5p/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        lr=d.get("long_run") or {}
        seg=[r["ms"] for r in (d.get("in_step_launches") or []) if "embed_grad_seg_kernel" in r["kernel"]]
        print(f.split("/")[-1], d["ms_per_step"], lr.get("mean_ms"), lr.get("p99_ms"), "hostmax", d.get("host_call_max_ms_in_window"), (d.get("host_stall") or {}).get("slowest_part_ms"), "seg", seg)
    except Exception as e: print(f, "ERR", e)
PY
grep . gpurun_out/r5p/probe_T*.log
