#!/bin/bash
# Matrix-core utilisation of the GEMM kernels of one bench configuration, from the SQ counters (north_star: "MFMA utilisation
# on the GEMMs against gfx950 peak"): one rocprofv3 --pmc pass (counters only, no tracing domain) over a short run, our GEMM
# kernels only.  SQ_VALU_MFMA_BUSY_CYCLES = the cycles the matrix pipes of all 1024 SIMDs were busy, summed (checked: 32 x the
# MFMA instruction count of a launch: the wide first layer at three products, 20.6 M x 32 = 6.6e8, measured 6.54e8);
# GRBM_GUI_ACTIVE = the active cycles of the 8 XCDs, summed (checked against the launch duration: 1.2e7 / 8 = 1.5e6 cycles =
# 0.75 ms at ~2.0 GHz).  util = busy / (GRBM_GUI_ACTIVE / 8 x 1024) per dispatch, averaged per kernel.
#   profiles/mfma_util.sh <tag> [bench.py flags...]      ->  profiles/<tag>_mfma_util.json (+ gpurun_out/profiles/)
set -u
export TMPDIR=/tmp
ulimit -c 0
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/prof_${TAG}_mfma
rm -rf $OUT; mkdir -p $OUT gpurun_out/profiles
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex 'linear_|cin_|mlp_tail|embed_gather_linear|embed_grad_|embed_ss_|embed_segsum' \
    --output-format csv -d $OUT -o m -- python bench.py --no-cpu-baseline --no-small-batch --long-steps 0 --pre-roll 0 --warmup 3 --steps 4 --graph off "$@" > $OUT/log.txt 2>&1
echo "mfma pass rc=$?"
python - "$OUT" "$TAG" <<'PY'
import csv, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
path = None
for root, _, files in os.walk(out):
    for f in files:
        if f.endswith("counter_collection.csv"):
            path = os.path.join(root, f)
if path is None:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].replace("void ", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    key = n[:100] + "|" + str(r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    b, g = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
    if not b or not g:
        continue
    bm, gm = sum(b) / len(b), sum(g) / len(g)
    res[k] = {"launches": len(b), "mfma_busy_cycles": bm, "gui_active_cycles": gm, "mfma_util": bm / (gm / 8.0 * 1024.0) if gm else None}
json.dump(res, open(f"profiles/{tag}_mfma_util.json", "w"), indent=1)
json.dump(res, open(f"gpurun_out/profiles/{tag}_mfma_util.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"])[:12]:
    print(f"{k[:90]:90s} n={v['launches']:4d} busy={v['mfma_busy_cycles']:.3e} active={v['gui_active_cycles']:.3e} util={v['mfma_util']:.3f}")
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
