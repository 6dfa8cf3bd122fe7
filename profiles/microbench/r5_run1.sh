set -x
mkdir -p gpurun_out/r5a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_kernels.py -k "embed_grad_seg or embed_grad_tiny" -x -q > gpurun_out/r5a/t_seg128.log 2>&1
RP_SEG_ROWS=64 timeout 600 python -m pytest tests/test_hip_kernels.py -k "embed_grad_seg" -x -q > gpurun_out/r5a/t_seg64.log 2>&1
RP_SEG_TILES=1 timeout 600 python -m pytest tests/test_hip_kernels.py -k "embed_grad_seg" -x -q > gpurun_out/r5a/t_seg128_T1.log 2>&1
timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5a/probe_128.log 2>&1
RP_SEG_ROWS=64 timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5a/probe_64.log 2>&1
for T in 2 4 16 32; do RP_SEG_TILES=$T timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5a/probe_128_T$T.log 2>&1; done
RP_SEG_ROWS=64 RP_SEG_TILES=16 timeout 300 python profiles/microbench/probes/probe_grad_seg.py > gpurun_out/r5a/probe_64_T16.log 2>&1
timeout 900 python -m pytest tests/test_hip_models.py tests/test_hip_graph.py -x -q -k "deepfm or graph" > gpurun_out/r5a/t_models.log 2>&1
timeout 600 python bench.py > gpurun_out/r5a/bench_seg.json 2> gpurun_out/r5a/bench_seg.err
RP_GRAD_SEG=0 timeout 600 python bench.py > gpurun_out/r5a/bench_old.json 2> gpurun_out/r5a/bench_old.err
tail -3 gpurun_out/r5a/t_*.log; cat gpurun_out/r5a/probe_128.log gpurun_out/r5a/probe_64.log; grep -h "round 5: embed_grad_seg (18" gpurun_out/r5a/probe_*T*.log
python - <<'PY'
import json
for n in ("seg","old"):
    try:
        d=json.loads(open(f"gpurun_out/r5a/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"])
    except Exception as e: print(n, "ERR", e)
PY
