import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from rec_pangu_amd import hip
from rec_pangu_amd.graph_step import GraphedTrainStep
from rec_pangu_amd.optim import make_adam
dev = torch.device("cuda")
enc = bench.criteo_enc_dict(1)
torch.manual_seed(0)
with torch.device(dev):
    model = bench.build_model("deepfm", enc)
for m in model.modules():
    if hasattr(m, "check_indices"):
        m.check_indices = "deferred"
model.train()
opt = make_adam(model, 1e-3)
B = 65536
g = GraphedTrainStep(model, opt)
gen = lambda i: bench.synth_batch(enc, B, 100 + i, dev)
nb = gen(0)
PRE = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for i in range(PRE):
    cur, nb = nb, gen(i + 1)
    g(cur, nb)
torch.cuda.synchronize()
print("after pre-roll: captures", g.captures, "replays", g.replays, "backend", g.backend_used,
      "| plan launches", g.plans[0].nodes, "side", g.plans[0].side, "inline", g.plans[0].inline)
batches = [gen(5000 + i) for i in range(20)]
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); hs = 0.0
    for i in range(20):
        th = time.perf_counter()
        g(batches[i % 20], batches[(i + 1) % 20])
        hs += time.perf_counter() - th
    e1.record(); torch.cuda.synchronize()
    print(f"rep {rep}: wall {(time.perf_counter()-t0)/20*1e3:.4f} ms/step, device (events) {e0.elapsed_time(e1)/20:.4f}, host {hs/20*1e3:.4f}, captures {g.captures}")
# the same batches eagerly
def eager(cur, nxt):
    model.prefetch(nxt); out = model(cur); out["loss"].backward(); opt.step(); model.zero_grad()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        eager(batches[i % 20], batches[(i + 1) % 20])
    torch.cuda.synchronize()
    print(f"eager rep {rep}: {(time.perf_counter()-t0)/20*1e3:.4f} ms/step")
