#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of one bench
# configuration, summarised into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json (copied to
# gpurun_out/profiles/ so that they travel back), then the bench JSON line of the same configuration (its
# `traffic` fields read the PMC summary just written).
#   profiles/collect.sh <tag> <pre-roll steps> [bench.py flags...]
# Counter collection serialises every dispatch it watches (~ms each), and bench.py's pre-roll is > 25 k dispatches: the
# PMC passes therefore watch ONLY our kernels (--kernel-include-regex) and ONLY their launches of the warm-up + timed
# steps (--kernel-iteration-range, one range per calls-per-step count); every profiler run is time-boxed.
set -u
export TMPDIR=/tmp
ulimit -c 0
cd "$(dirname "$0")/.."
TAG=$1; PRE=$2; shift 2
W=5; K=20
FLAGS="--no-cpu-baseline --no-small-batch --long-steps 0 --pre-roll $PRE --warmup $W --steps $K $*"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out/profiles
OURS='adam_kernel|embed_|linear_|wgrad_|transpose_kernel|relu_bwd|sigmoid_bce|loss_finish|zero_rows|iota_i32|cin_|crossnet|attn_|mmoe_|lazy_|counter_add|accumulate|pool_|fm_|bn_|batchnorm|field_sort|sort_hist|sort_scan|sort_scatter|mlp_tail|dropout|route_|shard_|DeviceRadixSort|radix|onesweep|multi_copy|copy_rows|dice_'
# steps before the timed region: (cold: W + min(K,64)) + PRE + W when PRE > 0, else W
if [ "$PRE" -gt 0 ]; then S0=$((W + K + PRE + W - 3)); else S0=$((W - 3)); fi  # (cold: min(K,64) = K at K=20)
S1=$((S0 + K + 3))
RANGES=""
for c in 1 2 3 4 5 6 8 10 12 14 16; do RANGES="$RANGES [$((S0 * c + 1))-$((S1 * c))]"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py $FLAGS > $OUT/stats.log 2>&1
echo "stats pass rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$OURS" --kernel-iteration-range $RANGES --output-format csv -d $OUT/fetch -o f -- python bench.py $FLAGS > $OUT/fetch.log 2>&1
echo "fetch pass rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$OURS" --kernel-iteration-range $RANGES --output-format csv -d $OUT/write -o w -- python bench.py $FLAGS > $OUT/write.log 2>&1
echo "write pass rc=$?"
python profiles/summarize.py $TAG $OUT/stats $OUT/fetch $OUT/write || { tail -5 $OUT/*.log; }
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json gpurun_out/profiles/ 2>/dev/null
wc -l $(find $OUT -name "*counter_collection.csv") 2>/dev/null
# the big raw traces stay on the box
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
timeout 400 python bench.py --pre-roll $PRE $* 2>/dev/null | grep "^{" > gpurun_out/profiles/${TAG}_bench.json
python - gpurun_out/profiles/${TAG}_bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"] or {}
print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r.get("kernel"), r.get("frac"), r.get("traffic"), (d.get("cold") or {}).get("ms_per_step"))
PY
