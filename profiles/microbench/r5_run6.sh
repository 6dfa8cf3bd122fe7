set -x
mkdir -p gpurun_out/r5f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_blocks.py -x -q -k "activation_epilogues or crossnet or cross" > gpurun_out/r5f/t_new.log 2>&1
timeout 900 python -m pytest tests/test_hip_models.py tests/test_hip_graph.py tests/test_hip_trainer.py -x -q -k "dcn or DCN or sharded_fused or stay_on_the_library" > gpurun_out/r5f/t_dcn.log 2>&1
timeout 300 python profiles/microbench/probes/aten_in_step.py dcn > gpurun_out/r5f/aten_dcn.log 2>&1
B="python bench.py --no-cpu-baseline --long-steps 600 --no-small-batch"
timeout 300 $B > gpurun_out/r5f/b_base.json 2>/dev/null
RP_PLAN_FORK=start timeout 300 $B > gpurun_out/r5f/b_forkstart.json 2>/dev/null
RP_SEG_TILES=4 timeout 300 $B > gpurun_out/r5f/b_T4.json 2>/dev/null
RP_SEG_TILES=16 timeout 300 $B > gpurun_out/r5f/b_T16.json 2>/dev/null
RP_GRAD_TINY=0 timeout 300 $B > gpurun_out/r5f/b_notiny.json 2>/dev/null
timeout 300 $B --model dcn > gpurun_out/r5f/b_dcn.json 2>/dev/null
for f in gpurun_out/r5f/t_*.log; do tail -n 2 $f; done
grep -v amdgpu gpurun_out/r5f/aten_dcn.log | tail -12
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5f/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        seg=[r["ms"] for r in (d.get("in_step_launches") or []) if r["kernel"].startswith(("embed_grad_seg_kernel","crossnet_bwd"))]
        print(f.split("/")[-1], d["ms_per_step"], d["long_run"]["mean_ms"], d["long_run"]["p99_ms"], seg, d["config"].get("captured_step_backend"))
    except Exception as e: print(f, "ERR", e)
PY
