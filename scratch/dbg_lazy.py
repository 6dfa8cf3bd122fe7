import torch, sys
sys.path.insert(0, '.')
from rec_pangu_amd import hip
DEV='cuda'
g = torch.Generator().manual_seed(0)
R, D, steps = 2000, 64, 6
p0 = torch.randn(R, D, generator=g)
b1, b2, eps = 0.9, 0.999, 1e-8
pd, md, vd = p0.to(DEV), torch.zeros(R, D, device=DEV), torch.zeros(R, D, device=DEV)
pl, ml, vl = p0.to(DEV), torch.zeros(R, D, device=DEV), torch.zeros(R, D, device=DEV)
last = torch.zeros(R, dtype=torch.int32, device=DEV)
table = torch.zeros(steps + 1, 2)
for t in range(1, steps + 1):
    lr = 1e-2
    table[t] = torch.tensor(hip.adam_step_scalars(lr, b1, b2, t))
    dev_table = table.to(DEV)
    rows = torch.randint(0, R // 2, (300,), generator=g)
    grad_rows = torch.randn(rows.numel(), D, generator=g)
    gd = torch.zeros(R, D).index_add_(0, rows, grad_rows).to(DEV)
    gl = gd.clone()
    hip.adam_step([pd.view(-1)], [gd.view(-1)], [md.view(-1)], [vd.view(-1)], lr, b1, b2, eps, t, zero_grad=True)
    sk, _ = hip.sort_pairs(rows.to(torch.int32).to(DEV), end_bit=13)
    hip.lazy_adam_rows(sk, D, pl, gl, ml, vl, last, dev_table, t, True, True, b1, b2, eps)
    cur = (last == t).nonzero().flatten()
    for name, a, b in (("p", pl, pd), ("m", ml, md), ("v", vl, vd)):
        d = (a[cur] - b[cur])
        print(t, name, "n_cur", cur.numel(), "ndiff", int((d != 0).sum()), "max", float(d.abs().max()), "rel", float((d.abs() / b[cur].abs().clamp(min=1e-30)).max()))
