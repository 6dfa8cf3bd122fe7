"""lockstep immediate vs deferred after a resume: first step at which the caught-up rows of the batch differ, and why"""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_hip_deferred_adam as T
from rec_pangu_amd import hip
from rec_pangu_amd.optim import make_adam

enc = T._enc(1, [500, 9, 4000])
batches = T._batches(enc, 128, 12, seed=3)
model = T._model("deepfm8", enc)
opt = make_adam(model, 1e-3, replay="exact", defer=True)
for i in range(6):
    model(batches[i])["loss"].backward(); opt.step(); model.zero_grad()
saved = (copy.deepcopy(model.state_dict()), copy.deepcopy(opt.state_dict()))
ms, os_ = [], []
for defer in (False, True):
    m = T._model("deepfm8", enc); o = make_adam(m, 1e-3, replay="exact", defer=defer)
    m.load_state_dict(saved[0]); o.load_state_dict(saved[1])
    ms.append(m); os_.append(o)
print("arena equal after load:", torch.equal(ms[0].embedding_layer.arena, ms[1].embedding_layer.arena))
for i in range(6, 10):
    b = batches[i]
    st = ms[0].embedding_layer
    keys = hip.embed_keys(st.row_base, st.row_count, [b[c].long().contiguous() for c in st.emb_feature], st.err_flag)
    u = torch.unique(keys).long()
    before = []
    for m in ms:
        lz = m.embedding_layer._lazy
        g = m.embedding_layer.grad_arena
        before.append((None if lz is None else lz.last[u].clone(), None if g is None else (g[u] != 0).any(1).clone(),
                       None if lz is None else (lz.t, lz._marked_for)))
    outs = [m(b) for m in ms]
    a0, a1 = ms[0].embedding_layer.arena[u], ms[1].embedding_layer.arena[u]
    bad = (a0 != a1).any(1)
    print(f"step {i+1}: pred equal {torch.equal(outs[0]['pred'], outs[1]['pred'])}; batch rows {u.numel()}, differing after catch-up {int(bad.sum())}; t/marked before: {before[0][2]} {before[1][2]}")
    if int(bad.sum()) and before[1][0] is not None:
        idx = bad.nonzero().flatten()[:10]
        print("   rows", u[idx].tolist()); print("   last imm ", before[0][0][idx].tolist()); print("   last def ", before[1][0][idx].tolist())
        print("   g nonzero def", before[1][1][idx].tolist(), " max |diff|", float((a0 - a1).abs().max()))
        lz0, lz1 = ms[0].embedding_layer._lazy, ms[1].embedding_layer._lazy
        print("   m equal", torch.equal(lz0.m[u[idx]], lz1.m[u[idx]]), " v equal", torch.equal(lz0.v[u[idx]], lz1.v[u[idx]]))
        print("   sc rows 6..9 imm", lz0.tabs.sc[6:10].tolist()); print("   sc rows 6..9 def", lz1.tabs.sc[6:10].tolist())
    for m, o, out in zip(ms, os_, outs):
        out["loss"].backward()
    ga, gb = ms[0].embedding_layer.grad_arena[u], ms[1].embedding_layer.grad_arena[u]
    print(f"   grads of the batch rows equal: {torch.equal(ga, gb)}")
    for m, o in zip(ms, os_):
        o.step(); m.zero_grad()
