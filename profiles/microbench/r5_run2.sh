set -x
mkdir -p gpurun_out/r5b
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_kernels.py -k "embed_grad_seg or embed_grad_tiny" -x -q > gpurun_out/r5b/t_seg128.log 2>&1
RP_SEG_ROWS=64 timeout 600 python -m pytest tests/test_hip_kernels.py -k "embed_grad_seg" -x -q > gpurun_out/r5b/t_seg64.log 2>&1
RP_SEG_TILES=1 timeout 600 python -m pytest tests/test_hip_kernels.py -k "embed_grad_seg" -x -q > gpurun_out/r5b/t_seg128_T1.log 2>&1
timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5b/probe_128.log 2>&1
RP_SEG_ROWS=64 timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5b/probe_64.log 2>&1
for T in 2 8 16; do RP_SEG_TILES=$T timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5b/probe_128_T$T.log 2>&1; done
RP_SEG_ROWS=64 RP_SEG_TILES=8 timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5b/probe_64_T8.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r5b/bench_seg.json 2> gpurun_out/r5b/bench_seg.err
for f in gpurun_out/r5b/t_*.log; do tail -n 2 $f; done
cat gpurun_out/r5b/probe_128.log; grep -H "round 5" gpurun_out/r5b/probe_*.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5b/bench_seg.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:10]:
    print("   %-45s calls %.1f mean %.4f" % (k, v["calls_per_step"], v["mean_ms"]))
PY
