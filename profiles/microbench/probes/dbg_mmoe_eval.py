import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), "tests"))
import torch
import test_hip_models as T
from rec_pangu_amd import functional as Fh
for name in ("mmoe_train", "mmoe_eval"):
    g = T.load_golden(f"model_{name}.npz")
    model = T.build(name).to("cuda")
    model.train(T.CASES[name][1])
    batch = T._to_dev(g["batch"])
    out = model(batch)
    print(name, {k: (float(v) if v.numel() == 1 else tuple(v.shape)) for k, v in out.items()}, "golden loss", float(g["out"]["loss"]))
    for i in (1, 2):
        p = out[f"task{i}_pred"].detach()
        gp = g["out"][f"task{i}_pred"]
        print("  pred diff", i, float((p.cpu() - gp).abs().max()))
        _, l = Fh.sigmoid_bce([p], batch[f"task{i}_label"].float(), apply_sigmoid=False, p_eps=1e-6, weight=0.5)
        print("  separate l", i, float(l))
