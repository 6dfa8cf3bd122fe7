import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch.multiprocessing as mp
import test_hip_sharded_world2 as T
import bench
from rec_pangu_amd.optim import make_adam
if __name__ == "__main__":
    mgr = mp.Manager(); ret = mgr.dict()
    T._spawn(T._worker, (2, T._free_port(), ret))
    r = [ret[0], ret[1]]
    enc = bench.criteo_enc_dict(T.SCALE)
    torch.manual_seed(1)
    with torch.device(T.DEV):
        plain = bench.build_model("deepfm", enc)
    emb = plain.embedding_layer
    with torch.no_grad():
        emb.arena.mul_(T.TABLE_SCALE)
    gb = T._global_batch(enc, 0)
    plain(gb)["loss"].backward(); plain.zero_grad()
    opt = make_adam(plain, 1e-2)
    seq = [T._global_batch(enc, 1 + i) for i in range(T.N_STEPS)] + [T._global_batch(enc, 50, T.BIG_B)]
    for i, b in enumerate(seq):
        o = plain(b); o["loss"].backward(); opt.step(); plain.zero_grad()
        got = torch.cat([r[0]["preds"][i], r[1]["preds"][i]])
        print("step", i, "pred diff", float((got - o["pred"].detach().cpu()).abs().max()))
    sd = plain.state_dict()
    for f, c in enumerate(emb.emb_feature):
        want = sd[f"embedding_layer.embedding_layer.{c}.weight"].cpu()
        got = r[0]["tables"][c]
        d = (got - want).abs()
        row = int(d.max(dim=1).values.argmax())
        touches = [(i, int((b[c].cpu() == row).sum()), int(((b[c].cpu() == row).nonzero().flatten() >= b[c].numel() // 2).sum())) for i, b in enumerate(seq)]
        print(c, want.shape[0], "maxdiff %.3e" % float(d.max()), "row", row, "owner", (int(emb.row_base[f]) + row) % 2, "touches(step,count,of which rank1)", [t for t in touches if t[1]])
    for k, v in r[0]["dense"].items():
        print(k, "dense diff %.3e" % float((v - sd[k].cpu()).abs().max()), "scale %.3e" % float(sd[k].abs().max()))
    # ---- control: the SAME arithmetic split without any exchange: gradient accumulation over the two half batches
    torch.manual_seed(1)
    with torch.device(T.DEV):
        acc = bench.build_model("deepfm", enc)
    with torch.no_grad():
        acc.embedding_layer.arena.mul_(T.TABLE_SCALE)
    from rec_pangu_amd.optim import FusedAdam
    opt2 = FusedAdam(acc.parameters(), lr=1e-2, fuse_zero_grad=True, lazy_tables=True)
    for i, b in enumerate(seq):
        n = b["label"].numel() // 2
        for h in range(2):
            hb = {k: v[h * n:(h + 1) * n].contiguous() for k, v in b.items()}
            (acc(hb)["loss"] * 0.5).backward()
        opt2.step(); acc.zero_grad()
    sd2 = acc.state_dict()
    print("control (grad accumulation over two halves, one process) vs plain / vs sharded:")
    for f, c in enumerate(emb.emb_feature):
        k = f"embedding_layer.embedding_layer.{c}.weight"
        print(c, "%.3e" % float((sd2[k] - sd[k]).abs().max()), "%.3e" % float((sd2[k].cpu() - r[0]["tables"][c]).abs().max()))
    for k, v in r[0]["dense"].items():
        print(k, "%.3e" % float((sd2[k] - sd[k]).abs().max()), "%.3e" % float((sd2[k].cpu() - v).abs().max()))
