"""lockstep: the same DeepFM trained with the serial and the closed-form replay; where do they part?"""
import sys, torch
sys.path.insert(0, ".")
from rec_pangu_amd.models.ranking import DeepFM
from rec_pangu_amd.optim import FusedAdam
DEV = "cuda"
enc = {f"I{i}": {"min": 0.0, "max": 1.0} for i in range(4)}
enc.update({f"C{i}": {"vocab_size": v} for i, v in enumerate([20000, 30, 7000, 3, 50000, 900])})
gen = torch.Generator().manual_seed(3)
B, NB = 256, 1101
big = {f"I{i}": torch.rand(NB, B, generator=gen).to(DEV) for i in range(4)}
big.update({f"C{i}": torch.randint(0, enc[f"C{i}"]["vocab_size"] + 1, (NB, B), generator=gen).to(DEV) for i in range(6)})
big["label"] = (torch.rand(NB, B, generator=gen) < 0.3).float().to(DEV)
batches = [{k: v[i].contiguous() for k, v in big.items()} for i in range(NB)]
ms, os_ = [], []
for replay in ("exact", "closed"):
    torch.manual_seed(0)
    m = DeepFM(embedding_dim=16, hidden_units=[32, 16], enc_dict=enc).to(DEV)
    ms.append(m); os_.append(FusedAdam(m.parameters(), lr=1e-3, fuse_zero_grad=True, lazy_tables=True, replay=replay))
for i in range(1100):
    outs = []
    for m, o in zip(ms, os_):
        out = m(batches[i]); out["loss"].backward(); o.step(); m.zero_grad(); outs.append(out["pred"].detach())
    if i >= 250 and (i < 300 or i % 50 == 0):
        la, lb = ms[0].embedding_layer, ms[1].embedding_layer
        cur = (la._lazy.last == la._lazy.t)
        assert torch.equal(la._lazy.last, lb._lazy.last)
        da = (la.arena[cur] - lb.arena[cur]).abs().max().item()
        dm = ((la._lazy.m[cur] - lb._lazy.m[cur]).abs().max() / la._lazy.m[cur].abs().max()).item()
        dv = ((la._lazy.v[cur] - lb._lazy.v[cur]).abs().max() / la._lazy.v[cur].abs().max()).item()
        dd = max((p - q).abs().max().item() for (n, p), (_, q) in zip(ms[0].named_parameters(), ms[1].named_parameters()) if "embedding" not in n)
        print(f"step {i + 1}: pred diff {(outs[0] - outs[1]).abs().max().item():.2e}  current rows p {da:.2e} m(rel) {dm:.2e} s(rel) {dv:.2e}  dense params {dd:.2e}")
