"""bisect 2: python probe_graph3.py <chunk> <eager_before 0/1> <lr_change 0/1>"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
chunk, eager_before, lrc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
from rec_pangu_amd.graph_step import GraphedTrainStep
from rec_pangu_amd.optim import FusedAdam, LazyAdamRows
from test_hip_graph import _enc, _batches, _build
LazyAdamRows.TABLE_CHUNK = chunk
enc = _enc(5, [3000, 17, 900, 4, 20000, 250])
steps = 60
bs = _batches(enc, 384, steps + 1, seed=4)
for mode in (["eager"] if eager_before else []) + ["graph"]:
    model = _build("deepfm64", enc)
    opt = FusedAdam(model.parameters(), lr=1e-3, fuse_zero_grad=True, lazy_tables=True, replay="closed")
    gs = GraphedTrainStep(model, opt) if mode == "graph" else None
    for i in range(steps):
        if lrc and i in (10, 40):
            for grp in opt.param_groups:
                grp["lr"] *= 0.5
        if gs is not None:
            out = gs(bs[i], bs[i + 1])
        else:
            model.prefetch(bs[i + 1]); out = model(bs[i]); out["loss"].backward(); opt.step(); model.zero_grad()
    torch.cuda.synchronize()
    sd = model.state_dict()
    print("done", mode, float(out["loss"]))
print("OK", chunk, eager_before, lrc)
