import sys, torch
sys.path.insert(0, ".")
import bench as B
from rec_pangu_amd import hip, functional as Fh
enc = B.criteo_enc_dict(1024)
dev = torch.device("cuda")
torch.manual_seed(0)
with torch.device(dev):
    model = B.build_model("deepfm", enc)
data = B.synth_batch(enc, 1024, 1, dev)
out = model(data)
lk = model.embedding_layer._fm_link
print("link", lk is not None, "ssum", None if lk is None else lk.ssum.shape)
out["loss"].backward()
print("dfm", None if lk.dfm is None else lk.dfm.shape, "folded", lk.folded)
x = torch.randn(1024, 1728, device=dev)
w = torch.randn(64, 1677, device=dev)
wt = hip.transpose(w, rows_out=1728)
a = torch.randn(1024, 64, device=dev)
out2 = torch.empty_like(x)
M, K = a.shape; N = wt.shape[0]
print(hip.get_matmul_precision(), K, M % 128, N % 64, lk.ncols, lk.ssum.shape, a.stride(0), wt.stride(0), a.data_ptr() % 16, wt.data_ptr() % 16, out2.shape)
print(hip.linear_fwd_rowadd(a, wt, lk.dfm.reshape(-1).contiguous(), lk.ssum, lk.ncols, out2))
