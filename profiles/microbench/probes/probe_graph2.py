"""bisect a capture crash: python probe_graph2.py <B> <vocab_big> <part>   part: sort | fwd | fwdbwd | step | gstep"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
B, V, part = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
from rec_pangu_amd import hip
from rec_pangu_amd.graph_step import GraphedTrainStep
from rec_pangu_amd.optim import FusedAdam
from rec_pangu_amd.models.layers import embedding as E
from test_hip_graph import _enc, _batches, _build
enc = _enc(5, [3000, 17, 900, 4, V, 250])
bs = _batches(enc, B, 8, seed=4)
model = _build("deepfm64", enc)
opt = FusedAdam(model.parameters(), lr=1e-3, fuse_zero_grad=True, lazy_tables=True, replay="closed")
if part == "gstep":
    gs = GraphedTrainStep(model, opt)
    for i in range(6):
        gs(bs[i], bs[i + 1])
    torch.cuda.synchronize(); print("OK gstep", B, V); sys.exit(0)
for i in range(3):
    model.prefetch(bs[i + 1]); model(bs[i])["loss"].backward(); opt.step(); model.zero_grad()
torch.cuda.synchronize()
lay = model.embedding_layer
g = torch.cuda.CUDAGraph()
if part == "sort":
    keys = hip.embed_keys(lay.row_base, lay.row_count, lay._idx_list(bs[4]), lay.err_flag)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        sk, sp = hip.sort_pairs(keys, end_bit=lay._meta()[3])
elif part == "fwd":
    with torch.cuda.graph(g):
        with torch.no_grad():
            out = model(bs[4], is_training=False)
elif part == "fwdbwd":
    with torch.cuda.graph(g):
        out = model(bs[4]); out["loss"].backward()
elif part == "step":
    opt.prepare_step(); opt.set_device_clock(True)
    with torch.cuda.graph(g):
        out = model(bs[4]); out["loss"].backward(); opt.step(); model.zero_grad()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
print("OK", part, B, V)
