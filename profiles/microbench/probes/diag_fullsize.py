"""Which table-gradient rows differ between the HIP DeepFM and the CPU oracle at B = 65536 (bench.full_size_parity)?
Prints, per execution form of the first layer's backward (RP_GRAD_SEG = 1 / 0), the worst tables and rows with their run
lengths, and the two HIP forms against each other."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from rec_pangu_amd.models.ranking import DeepFM  # noqa: E402

scale = int(os.environ.get("DIAG_SCALE", "64"))
leg = bench.oracle_first_step(scale=scale)
enc, first = leg["enc"], leg["first"]
dev = torch.device("cuda")
batch = {k: v.to(dev) for k, v in leg["batch"].items()}
grads = {}
for mode in ("1", "0"):
    os.environ["RP_GRAD_SEG"] = mode
    with torch.device(dev):
        m = DeepFM(embedding_dim=64, hidden_units=[64, 64, 64], enc_dict=enc)
    m.load_state_dict(leg["state0"])
    m.train()
    out = m(batch)
    out["loss"].backward()
    torch.cuda.synchronize()
    grads[mode] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
    print(f"== RP_GRAD_SEG={mode}: pred diff {float((out['pred'].detach().cpu() - first['pred']).abs().max()):.2e}")
    rows = []
    for k, g in grads[mode].items():
        ref = first["grads"][k]
        rows.append((float((g - ref).abs().max()) / max(float(ref.abs().max()), 1e-12), k))
    rows.sort(reverse=True)
    for e, k in rows[:6]:
        print(f"   {k:55s} err/scale {e:.3e}  max|ref| {float(first['grads'][k].abs().max()):.3e}")
    e, k = rows[0]
    if "embedding_layer" in k:
        col = k.split(".")[2]
        g, ref = grads[mode][k], first["grads"][k]
        d = (g - ref).abs().amax(dim=1)
        top = torch.topk(d, 5)
        ids = leg["batch"][col]
        cnts = torch.bincount(ids, minlength=g.shape[0])
        for v, r in zip(top.values.tolist(), top.indices.tolist()):
            print(f"      row {r}: |diff| {v:.3e}  |ref| {float(ref[r].abs().max()):.3e}  lookups {int(cnts[r])}  "
                  f"hip {g[r, :3].tolist()}  ref {ref[r, :3].tolist()}")
        nz = (d > 1e-3 * float(ref.abs().max())).sum()
        print(f"      rows off by more than 1e-3 of scale: {int(nz)} of {int((cnts > 0).sum())} looked up")
for k in grads["1"]:
    e = float((grads["1"][k] - grads["0"][k]).abs().max()) / max(float(grads["0"][k].abs().max()), 1e-12)
    if e > 1e-5:
        print(f"seg vs pair form: {k} {e:.3e}")
print("done")
