cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2c
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2c/prof -o p -- python profiles/microbench/probes/probe_grad_ss.py > gpurun_out/s2c/probe.txt 2>&1
cat gpurun_out/s2c/probe.txt | grep -v "^W20\|rocprof" | tail -12
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/s2c/prof/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find gpurun_out/s2c -name "*.csv" -size +1M -delete; find gpurun_out/s2c -name "*.db" -delete
