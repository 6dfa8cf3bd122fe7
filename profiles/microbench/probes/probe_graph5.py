"""reproduce the full-size graphed run quickly: python probe_graph5.py <B> <steps>"""
import sys, time, torch
sys.path.insert(0, ".")
import bench as Bn
from rec_pangu_amd.optim import make_adam
from rec_pangu_amd.graph_step import GraphedTrainStep
dev = torch.device("cuda:0")
B, steps = int(sys.argv[1]), int(sys.argv[2])
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
bs = [Bn.synth_batch(enc, B, 100 + i, dev, "uniform") for i in range(32)]
t0 = time.perf_counter()
for i in range(steps):
    gs(bs[i % 32], bs[(i + 1) % 32])
    if i % 50 == 49:
        torch.cuda.synchronize(); print("step", i + 1, "ok", flush=True)
torch.cuda.synchronize()
t1 = time.perf_counter()
for i in range(steps, steps + 40):
    gs(bs[i % 32], bs[(i + 1) % 32])
torch.cuda.synchronize()
print(f"OK B={B}: {(time.perf_counter() - t1) / 40 * 1e3:.3f} ms/step")
model.embedding_layer.raise_if_bad_index()
