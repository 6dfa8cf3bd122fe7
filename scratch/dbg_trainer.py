import os, sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import GOLDEN
from test_trainer_dataset import _loaders_in_reference_order
from rec_pangu_amd import hip
from rec_pangu_amd.models.ranking import DeepFM
from rec_pangu_amd.optim import make_adam
meta, train_loader, valid_loader, test_loader, enc, test_df = _loaders_in_reference_order()
torch.manual_seed(0)
model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc).to("cuda")
opt = make_adam(model, 1e-3)
dev = torch.device("cuda")
for ep in range(2):
    for i, data in enumerate(train_loader):
        data = {k: v.to(dev) for k, v in data.items()}
        print("train", ep, i, data["label"].shape[0], flush=True)
        out = model(data); torch.cuda.synchronize(); print(" fwd ok", flush=True)
        out["loss"].backward(); torch.cuda.synchronize(); print(" bwd ok", flush=True)
        opt.step(); torch.cuda.synchronize(); print(" step ok", flush=True)
        model.zero_grad()
    model.eval()
    with torch.no_grad():
        for i, data in enumerate(valid_loader):
            data = {k: v.to(dev) for k, v in data.items()}
            print("eval", i, data["label"].shape[0], flush=True)
            out = model(data, is_training=False); torch.cuda.synchronize()
    model.train()
    print("state_dict", flush=True)
    sd = model.state_dict(); torch.cuda.synchronize()
    torch.save({"model": sd}, "/tmp/x.pth"); print(" saved", flush=True)
print("done")
