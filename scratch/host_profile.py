"""Scratch: cProfile of the host side of one DeepFM train step (tiny batch -> GPU time negligible)."""
import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
import bench as B
from rec_pangu_amd import hip
from rec_pangu_amd.optim import make_adam
enc = B.criteo_enc_dict(64)
dev = torch.device("cuda")
torch.manual_seed(0)
with torch.device(dev):
    model = B.build_model("deepfm", enc)
for m in model.modules():
    if hasattr(m, "check_indices"): m.check_indices = "deferred"
opt = make_adam(model, 1e-3)
data = B.synth_batch(enc, 256, 1, dev)
def step():
    out = model(data); out["loss"].backward(); opt.step(); model.zero_grad()
for _ in range(20): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
