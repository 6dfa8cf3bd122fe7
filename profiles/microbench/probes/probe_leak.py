"""Does a replayed step keep the batches it was given alive?  memory_allocated over 1500 replays on fresh batches."""
import os, sys, time, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
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
g = GraphedTrainStep(model, opt)
gen = lambda i: bench.synth_batch(enc, 65536, 100 + i, dev)
nb = gen(0)
for i in range(1500):
    cur, nb = nb, gen(i + 1)
    g(cur, nb)
    if i % 250 == 0:
        torch.cuda.synchronize()
        print(i, "allocated GB", round(torch.cuda.memory_allocated() / 2**30, 3), "reserved", round(torch.cuda.memory_reserved() / 2**30, 3), flush=True)
del cur, nb
torch.cuda.synchronize()
# lists of fresh batches, dropped
for rep in range(4):
    pw = [gen(900000 + rep * 1000 + i) for i in range(256)]
    for i in range(255):
        g(pw[i], pw[i + 1])
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_allocated()
    del pw
    gc.collect()
    torch.cuda.synchronize()
    print("rep", rep, "allocated GB before / after del", round(a0 / 2**30, 3), round(torch.cuda.memory_allocated() / 2**30, 3), flush=True)
