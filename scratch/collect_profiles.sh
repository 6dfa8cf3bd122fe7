#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of the
# default bench command, then the bench JSON lines of every configuration.  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=${1:-r01}
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o s -- $BENCH > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o f -- $BENCH > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o w -- $BENCH > gpurun_out/prof_write.log 2>&1
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" | head
SD=$(dirname $(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1))
FD=$(dirname $(find gpurun_out/prof_fetch -name "*counter_collection.csv" | head -1))
WD=$(dirname $(find gpurun_out/prof_write -name "*counter_collection.csv" | head -1))
python profiles/summarize.py $TAG $SD $FD $WD && mkdir -p gpurun_out/profiles && cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json gpurun_out/profiles/
# bench lines (the PMC summary above is now in profiles/ on this box, so `traffic` is filled in)
python bench.py 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench_n1_deepfm.json
python bench.py --mode forward --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench_n1_deepfm_fwd.json
python bench.py --hidden 1024,512,256 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench_n1_deepfm_wide.json
python bench.py --optimizer dense --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench_n1_dense.json
for m in xdeepfm dcn autoint mmoe; do
  python bench.py --model $m --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench_n1_$m.json
done
python bench.py --sharded --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench_n1_deepfm_sharded_path.json
for f in gpurun_out/profiles/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"] or {}
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r.get("kernel"), r.get("frac"), (d.get("cpu_baseline") or {}).get("value"))
PY
done
