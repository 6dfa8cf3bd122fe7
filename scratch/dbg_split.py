import sys, torch
sys.path.insert(0, '.')
import bench
DEV = 'cuda'
enc = bench.criteo_enc_dict(16)
torch.manual_seed(7)
model = bench.build_model('xdeepfm', enc).to(DEV)
model.train()
B = 65536
full = bench.synth_batch(enc, B, 11, DEV)
h0 = {k: v[:B // 2] for k, v in full.items()}
h1 = {k: v[B // 2:] for k, v in full.items()}
with torch.no_grad():
    pf = model(full)['pred']; p0 = model(h0)['pred']; p1 = model(h1)['pred']
    pf2 = model(full)['pred']
d = (pf - torch.cat([p0, p1])).abs().reshape(-1)
print('rerun diff', float((pf - pf2).abs().max()))
print('max diff', float(d.max()), 'n>1e-5', int((d > 1e-5).sum()), 'first idx', (d > 1e-5).nonzero()[:10].reshape(-1).tolist())
print('pred range', float(pf.min()), float(pf.max()))
# CIN alone
cin = model.cin
x, _ = model.embedding_layer.gather_concat(full, [], want_fm=False)
F, D = 26, 64
with torch.no_grad():
    yf = cin(x[:, :F * D].unflatten(1, (F, D)))
    y0 = cin(x[:B // 2, :F * D].unflatten(1, (F, D)))
    y1 = cin(x[B // 2:, :F * D].unflatten(1, (F, D)))
dc = (yf - torch.cat([y0, y1])).abs().reshape(-1)
print('cin max diff', float(dc.max()), 'scale', float(yf.abs().max()), 'bad', (dc > 1e-4 * float(yf.abs().max())).nonzero()[:10].reshape(-1).tolist())
