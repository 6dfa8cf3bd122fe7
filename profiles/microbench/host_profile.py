"""Where the host spends its time while enqueueing a DeepFM train step (cProfile over 200 steps, Criteo shape):
python profiles/microbench/host_profile.py"""
import cProfile
import pstats
import sys
import torch
sys.path.insert(0, '.')
import bench
from rec_pangu_amd.optim import make_adam

dev = torch.device("cuda:0")
enc = bench.criteo_enc_dict(1)
torch.manual_seed(0)
model = bench.build_model("deepfm", enc).to(dev)
for m in model.modules():
    if hasattr(m, "check_indices"):
        m.check_indices = "deferred"
opt = make_adam(model, 1e-3)
batches = [bench.synth_batch(enc, 65536, 100 + i, dev) for i in range(32)]


def step(i):
    model.prefetch(batches[(i + 1) % 32])
    out = model(batches[i % 32])
    out["loss"].backward()
    opt.step()
    model.zero_grad()


for i in range(20):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
