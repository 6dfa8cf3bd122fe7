ulimit -c 0
export TMPDIR=/tmp
mkdir -p gpurun_out/r5ae gpurun_out/profiles
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r5ae/pytest_full.txt 2>&1
grep -n "passed\|failed\|Error\|^E " gpurun_out/r5ae/pytest_full.txt | head -20
run() { tag=$1; shift; timeout 170 python bench.py --no-cpu-baseline --no-small-batch --long-steps 300 "$@" 2>/dev/null | grep "^{" > gpurun_out/profiles/r05_bench_$tag.json
  python - gpurun_out/profiles/r05_bench_$tag.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"] or {}
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], (d.get("long_run") or {}).get("mean_ms"), d["config"].get("captured_step_backend"), r.get("kernel"), r.get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run mmoe --model mmoe
run autoint --model autoint
run dcn --model dcn
# rocprofv3 kernel statistics of the secondary configurations (round 5)
for m in mmoe xdeepfm autoint dcn; do
  rm -rf gpurun_out/prof_$m; mkdir -p gpurun_out/prof_$m
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -o s -- python bench.py --model $m --no-cpu-baseline --no-small-batch --long-steps 0 --pre-roll 200 > gpurun_out/prof_$m/log.txt 2>&1
  f=$(find gpurun_out/prof_$m -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -25 "$f" > gpurun_out/profiles/r05_${m}_kernel_stats.csv; fi
  find gpurun_out/prof_$m -name "*.csv" -size +1M -delete; find gpurun_out/prof_$m -name "*.db" -delete
  head -4 gpurun_out/profiles/r05_${m}_kernel_stats.csv | cut -c1-160
done
