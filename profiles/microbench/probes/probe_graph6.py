"""stress the graphed step the way bench.py drives it: python probe_graph6.py <B> <steps> <fresh 0/1> <interleave 0/1> <sync_every>"""
import sys, time, torch
sys.path.insert(0, ".")
import bench as Bn
from rec_pangu_amd.optim import make_adam
from rec_pangu_amd.graph_step import GraphedTrainStep
dev = torch.device("cuda:0")
B, steps, fresh, inter, sync_every = [int(x) for x in sys.argv[1:6]]
enc = Bn.criteo_enc_dict(1)
torch.manual_seed(0)
with torch.device(dev):
    model = Bn.build_model("deepfm", enc, (64, 64, 64))
for m in model.modules():
    if hasattr(m, "check_indices"):
        m.check_indices = "deferred"
model.train()
opt = make_adam(model, 1e-3)
gs = GraphedTrainStep(model, opt)
pool = [Bn.synth_batch(enc, B, 100 + i, dev, "uniform") for i in range(32)]
gen = (lambda i: Bn.synth_batch(enc, B, 100 + i, dev, "uniform")) if fresh else (lambda i: pool[i % 32])
def eager(a, b):
    model.prefetch(b); out = model(a); out["loss"].backward(); opt.step(); model.zero_grad()
nb = gen(0)
t_start = None
for i in range(steps):
    if i == 100:
        torch.cuda.synchronize(); t_start = time.perf_counter()
    cur, nb = nb, gen(i + 1)
    if inter and i % 400 in (395, 396, 397):
        eager(cur, nb)
    else:
        gs(cur, nb)
    if sync_every and i % sync_every == sync_every - 1:
        torch.cuda.synchronize()
    if i % 200 == 199:
        print("enqueued", i + 1, "replays", gs.replays, flush=True)
torch.cuda.synchronize()
if t_start is not None:
    print(f"ms/step over the last {steps - 100} steps: {(time.perf_counter() - t_start) / (steps - 100) * 1e3:.4f}")
model.embedding_layer.raise_if_bad_index()
print(f"OK B={B} steps={steps} fresh={fresh} inter={inter} sync={sync_every} replays={gs.replays} t={model.embedding_layer._lazy.t}")
