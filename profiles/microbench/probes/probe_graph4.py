"""where does the host time of a graphed step go?"""
import sys, time, torch
sys.path.insert(0, ".")
import bench as Bn
from rec_pangu_amd.optim import make_adam
from rec_pangu_amd.graph_step import GraphedTrainStep
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
gs = GraphedTrainStep(model, opt)
bs = [Bn.synth_batch(enc, B, 100 + i, dev, "uniform") for i in range(64)]
for i in range(40):
    gs(bs[i], bs[i + 1])
torch.cuda.synchronize()
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
gs._copy = timed("copy", gs._copy)
gs._fits = timed("fits", gs._fits)
opt.prepare_step = timed("prepare", opt.prepare_step)
opt.host_counters = timed("counters", opt.host_counters)
opt.advance_host = timed("advance", opt.advance_host)
for P in (0, 1):
    g = gs.graphs[P]
    class W:
        def __init__(s, g): s.g = g
        def replay(s):
            t = time.perf_counter(); s.g.replay(); acc["replay"] = acc.get("replay", 0.0) + time.perf_counter() - t
    gs.graphs[P] = W(g)
t0 = time.perf_counter(); th = 0.0
N = 20
for i in range(40, 40 + N):
    t1 = time.perf_counter(); gs(bs[i % 64], bs[(i + 1) % 64]); th += time.perf_counter() - t1
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step, host {th / N * 1e3:.3f} ms/step;", {k: round(v / N * 1e3, 4) for k, v in acc.items()})
