"""diagnostic: device closed-form table and one replay launch against a numpy float64 emulation"""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from rec_pangu_amd import hip
DEV = "cuda"
b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
steps, CF = 1000, 256
table = torch.zeros(steps + 1, 2); nsd = torch.zeros(steps + 1, 2, dtype=torch.float64)
for t in range(1, steps + 1):
    table[t] = torch.tensor(hip.adam_step_scalars(lr, b1, b2, t, eps))
    nsd[t, 0] = -lr / (1 - b1 ** t); nsd[t, 1] = 1 / (1 - b2 ** t) ** 0.5
cf = torch.zeros(steps + 8, 8, device=DEV)
hip.lazy_adam_cf_table(nsd.to(DEV), steps, CF, b1, b2, cf)  # (built_to=-1: a full build)
torch.cuda.synchronize()
cfh = cf.cpu().numpy()
ns = nsd[:, 0].numpy(); d = nsd[:, 1].numpy()
b1e = 1.0 - float(np.float32(1.0 - b1)); r = float(np.float32(np.sqrt(b2)))
def entry(t_end, k, J=372):
    l = t_end - k; n = min(k, J); i = np.arange(1, n + 1)
    w = ns[l + i] * np.exp(np.log(b1e) * i); a = d[l + i] * np.exp(np.log(r) * i)
    sw = w.sum(); abar = (w * a).sum() / sw; da = a - abar
    return np.array([abar, sw, (w * da ** 2).sum(), -(w * da ** 3).sum(), (w * da ** 4).sum(), b1e ** k, r ** k, 0])
for k in (1, 2, 5, 63, 64, 65, 200, 372, 373, 500, 744):
    e = entry(steps, k)
    print(k, "dev", cfh[k], "\n   ref", e.astype(np.float32), "maxrel", np.abs(cfh[k][:7] / e[:7] - 1).max())
# one replay launch
for D in (64, 16):
    R = 800
    g = torch.Generator().manual_seed(0)
    p = 0.05 * torch.randn(R, D, generator=g); s = 10.0 ** (-7 + 5 * torch.rand(R, D, generator=g)); m = s * torch.randn(R, D, generator=g)
    last = torch.randint(1, steps, (R,), generator=g).to(torch.int32)
    keys = torch.arange(R, dtype=torch.int32)
    out = {}
    for kind in ("serial", "closed"):
        P, M, S, L = p.to(DEV), m.to(DEV), s.to(DEV), last.to(DEV)
        hip.lazy_adam_rows(keys.to(DEV), D, P, None, M, S, L, table.to(DEV), steps, False, False, b1, b2, eps,
                           cf if kind == "closed" else None, CF if kind == "closed" else 0)
        out[kind] = (P.cpu(), M.cpu(), S.cpu())
    dp = (out["closed"][0] - out["serial"][0]).abs()
    upd = (out["serial"][0] - p).abs()
    rel = dp / upd.clamp_min(1e-12)
    i = int(rel.amax(1).argmax())
    print(f"D={D}: max |dp| {float(dp.max()):.3e}; max rel-to-update {float(rel.max()):.3e} at row {i} last {int(last[i])}; "
          f"m rel {float(((out['closed'][1]-out['serial'][1]).abs()/out['serial'][1].abs().clamp_min(1e-30)).max()):.2e} "
          f"s rel {float(((out['closed'][2]-out['serial'][2]).abs()/out['serial'][2].abs().clamp_min(1e-30)).max()):.2e}")
    worst = rel.amax(1)
    for lo, hi in ((1, 256), (256, 600), (600, 936), (936, 1000)):
        sel = (last >= lo) & (last < hi)
        if sel.any():
            print(f"   last in [{lo},{hi}): rows {int(sel.sum())}, max rel {float(worst[sel].max()):.3e}")

# float64 emulation of the serial replay with the kernels' own constants, per element
print("---- against a float64 emulation of the zero-gradient steps (kernel constants) ----")
A = table[:, 0].double().numpy(); Bc = table[:, 1].double().numpy()
for D in (64,):
    R = 800
    g = torch.Generator().manual_seed(0)
    p = 0.05 * torch.randn(R, D, generator=g); s = 10.0 ** (-7 + 5 * torch.rand(R, D, generator=g)); m = s * torch.randn(R, D, generator=g)
    last = torch.randint(1, steps, (R,), generator=g).to(torch.int32)
    keys = torch.arange(R, dtype=torch.int32)
    P64, M64, S64 = p.double().numpy().copy(), m.double().numpy().copy(), s.double().numpy().copy()
    ln = last.numpy()
    for j in range(2, steps + 1):
        act = (ln < j)[:, None]
        M2 = M64 * b1e; S2 = S64 * r
        P2 = P64 + M2 / (S2 * A[j] + Bc[j])
        M64 = np.where(act, M2, M64); S64 = np.where(act, S2, S64); P64 = np.where(act, P2, P64)
    out = {}
    for kind in ("serial", "closed"):
        Pd, Md, Sd, Ld = p.to(DEV), m.to(DEV), s.to(DEV), last.to(DEV)
        hip.lazy_adam_rows(keys.to(DEV), D, Pd, None, Md, Sd, Ld, table.to(DEV), steps, False, False, b1, b2, eps,
                           cf if kind == "closed" else None, CF if kind == "closed" else 0)
        e = (Pd.cpu().double().numpy() - P64)
        ulp = np.spacing(np.abs(p.numpy()).astype(np.float32)).astype(np.float64)
        print(kind, "p err abs max %.3e rms %.3e | in ulps of p: max %.1f rms %.2f mean(signed) %.2f" %
              (np.abs(e).max(), np.sqrt((e ** 2).mean()), np.abs(e / ulp).max(), np.sqrt(((e / ulp) ** 2).mean()), (e / ulp * np.sign(m.numpy())).mean()))
        for lo, hi in ((1, 256), (256, 600), (600, 936), (936, 1000)):
            sel = (ln >= lo) & (ln < hi)
            print("    last in [%d,%d): rms ulps %.2f  mean signed (along m) %.2f" % (lo, hi, np.sqrt(((e / ulp)[sel] ** 2).mean()), ((e / ulp) * np.sign(m.numpy()))[sel].mean()))
        if kind == "serial":
            keep_serial = Pd.cpu().double().numpy()
    upd = P64 - p.double().numpy()
    big = np.abs(upd) > 1e-4
    for kind, arr in (("serial", keep_serial), ("closed", Pd.cpu().double().numpy())):
        rel = (arr - P64) / upd
        print(kind, "relative to the true update (|update| > 1e-4, %d elements):" % big.sum())
        for lo, hi in ((1, 256), (256, 400), (400, 600), (600, 936), (936, 990), (990, 1000)):
            sel = ((ln >= lo) & (ln < hi))[:, None] & big
            if sel.any():
                print("    last in [%d,%d): mean %.3e rms %.3e max %.3e" % (lo, hi, rel[sel].mean(), np.sqrt((rel[sel] ** 2).mean()), np.abs(rel[sel]).max()))
    # the closed formula evaluated in float64 with the DEVICE table, and in float32 step by step
    tab = cfh.astype(np.float64)
    k = (steps - np.maximum(ln, CF))
    e = tab[k]  # [R, 8]
    s0, m0, p0 = s.double().numpy(), m.double().numpy(), p.double().numpy()
    # rows below CF: skip in this check
    rows = ln >= CF
    q = 1 / (s0 * e[:, 0:1] + eps); y = s0 * q
    poly = e[:, 1:2] + y * y * (e[:, 2:3] + y * (e[:, 3:4] + y * e[:, 4:5]))
    pc64 = p0 + m0 * q * poly
    rel = ((pc64 - P64) / upd)
    sel = rows[:, None] & big
    print("closed formula in float64 with the device table vs emulation: mean %.3e rms %.3e max %.3e" % (rel[sel].mean(), np.sqrt((rel[sel] ** 2).mean()), np.abs(rel[sel]).max()))
