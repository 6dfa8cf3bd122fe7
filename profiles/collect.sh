#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of one bench
# configuration, summarised into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json (copied to
# gpurun_out/profiles/ so that they travel back), then the bench JSON line of the same configuration (its
# `traffic` fields read the PMC summary just written).
#   profiles/collect.sh <tag> [bench.py flags...]
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
TAG=$1; shift
BENCH="python bench.py --no-cpu-baseline $*"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out/profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $BENCH > $OUT/write.log 2>&1
python profiles/summarize.py $TAG $OUT/stats $OUT/fetch $OUT/write || { tail -5 $OUT/*.log; exit 1; }
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json gpurun_out/profiles/
# the big raw traces stay on the box
find $OUT -name "*.csv" -size +2M -delete
python bench.py $* 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench.json
python - gpurun_out/profiles/${TAG}_bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"] or {}
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r.get("kernel"), r.get("frac"), r.get("traffic"), (d.get("cold") or {}).get("ms_per_step"))
PY
