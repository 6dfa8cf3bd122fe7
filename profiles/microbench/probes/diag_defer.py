"""why does a RESUMED run differ between immediate and deferred execution?  prints where the two runs part"""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_hip_deferred_adam as T
from rec_pangu_amd.optim import make_adam

enc = T._enc(1, [500, 9, 4000])
batches = T._batches(enc, 128, 12, seed=3)
model = T._model("deepfm8", enc)
opt = make_adam(model, 1e-3, replay="exact", defer=True)
for i in range(6):
    model(batches[i])["loss"].backward(); opt.step(); model.zero_grad()
saved = (copy.deepcopy(model.state_dict()), copy.deepcopy(opt.state_dict()))
print("saved groups:", [{k: v for k, v in g.items() if k != "params"} for g in saved[1]["param_groups"]])
for variant in ("resume", "scratch"):
    runs = []
    for defer in (False, True):
        model = T._model("deepfm8", enc)
        opt = make_adam(model, 1e-3, replay="exact", defer=defer)
        if variant == "resume":
            model.load_state_dict(saved[0]); opt.load_state_dict(saved[1])
        trace = []
        for i in range(6, 12):
            out = model(batches[i]); out["loss"].backward(); opt.step(); model.zero_grad()
            lz = model.embedding_layer._lazy
            trace.append((out["pred"].detach().clone(), lz.t, lz._marked_for, int((lz.last < 0).sum()), int((lz.last > 0).sum())))
        runs.append((trace, T._state(model, opt), model))
    (ta, sa, ma), (tb, sb, mb) = runs
    for i, (a, b) in enumerate(zip(ta, tb)):
        print(variant, "step", i + 7, "pred equal", torch.equal(a[0], b[0]), "max diff", float((a[0] - b[0]).abs().max()),
              "| imm t/marked/neg/pos", a[1:], "| def", b[1:])
    for k in sa:
        if not torch.equal(sa[k], sb[k]):
            d = (sa[k].float() - sb[k].float()).abs()
            rows = (d.reshape(d.shape[0], -1).max(1).values > 0).nonzero().flatten() if d.dim() > 0 and d.shape[0] > 0 else []
            print(variant, "DIFF", k, tuple(sa[k].shape), "max", float(d.max()), "rows differing", len(rows), rows[:8].tolist() if len(rows) else "")
    print(variant, "done")
