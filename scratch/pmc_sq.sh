#!/bin/bash
# SQ stall breakdown per kernel (one PMC pass, no tracing): usage pmc_sq.sh <bench args...>
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --output-format csv -d gpurun_out/pmc_sq -o q -- python bench.py --no-cpu-baseline --steps 3 --warmup 2 "$@" > gpurun_out/pmc_sq.log 2>&1
F=$(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
python - "$F" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": cnt[k]+=1
rows=sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_BUSY_CYCLES",0))
print(f'{"kernel":60s} {"n":>4s} {"wave_cyc(M)":>11s} {"wait_any":>8s} {"wait_inst":>9s} {"active":>7s} {"w_lds":>6s} {"mfma_busy/busy":>14s} {"lds_conf/wave":>13s}')
for k,v in rows[:22]:
    wc=v.get("SQ_WAVE_CYCLES",1) or 1
    print(f'{k:60s} {cnt[k]:4d} {wc/1e6:11.1f} {v.get("SQ_WAIT_ANY",0)/wc:8.2f} {v.get("SQ_WAIT_INST_ANY",0)/wc:9.2f} {v.get("SQ_ACTIVE_INST_ANY",0)/wc:7.2f} {v.get("SQ_WAIT_INST_LDS",0)/wc:6.2f} {v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(v.get("SQ_BUSY_CYCLES",1),1):14.3f} {v.get("SQ_LDS_BANK_CONFLICT",0)/wc:13.3f}')
PY
