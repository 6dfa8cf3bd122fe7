"""probe: does a whole train step (fwd + bwd + FusedAdam, sort-ahead on the side stream) capture into a hipGraph as it
is, and what does a replay cost on the device / the host?  (step counters are baked in: timing only, not a valid run)"""
import sys, time, torch
sys.path.insert(0, ".")
import bench as Bn
from rec_pangu_amd import hip
from rec_pangu_amd.optim import make_adam
from rec_pangu_amd.models.layers import embedding as E
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
enc = Bn.criteo_enc_dict(1)
torch.manual_seed(0)
with torch.device(dev):
    model = Bn.build_model("deepfm", enc, (64, 64, 64))
for m in model.modules():
    if hasattr(m, "check_indices"):
        m.check_indices = "deferred"
model.train()
opt = make_adam(model, 1e-3)
gen = lambda i: Bn.synth_batch(enc, B, 100 + i, dev, "uniform")

def step(data, nxt=None):
    if nxt is not None:
        model.prefetch(nxt)
    out = model(data)
    out["loss"].backward()
    opt.step()
    model.zero_grad()

bs = [gen(i) for i in range(300)]
for i in range(299):
    step(bs[i], bs[i + 1])
torch.cuda.synchronize()
t0 = time.perf_counter(); th = 0.0
for i in range(100, 150):
    t1 = time.perf_counter(); step(bs[i], bs[i + 1]); th += time.perf_counter() - t1
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step, host {th / 50 * 1e3:.3f} ms/step")
cur, nxt = bs[200], bs[201]
model.prefetch(cur); step(bs[199], None)  # cur's sort is in the cache now
torch.cuda.synchronize()
for e in E._SORT_CACHE:
    e[4] = None
side = E._SIDE_STREAMS[dev]
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        model.prefetch(nxt)
        out = model(cur)
        out["loss"].backward()
        opt.step()
        model.zero_grad()
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("captured")
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); th = 0.0
    for _ in range(50):
        t1 = time.perf_counter(); g.replay(); th += time.perf_counter() - t1
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step, host {th / 50 * 1e3:.3f} ms/step")
except Exception as ex:
    import traceback; traceback.print_exc()
    print("CAPTURE FAILED:", repr(ex)[:400])
