mkdir -p gpurun_out/s2e
python - > gpurun_out/s2e/fullvocab.txt 2>&1 <<'PY'
import torch, bench, json, time
t=time.time()
leg = bench.oracle_first_step(scale=1, adam=False)
print("oracle s", time.time()-t)
res = bench.full_size_parity(leg, torch.device("cuda"))
print(json.dumps(res, indent=1))
print("total s", time.time()-t)
PY
tail -30 gpurun_out/s2e/fullvocab.txt
