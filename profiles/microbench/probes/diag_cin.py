"""CIN block gradient errors against the float64 oracle, per unit configuration and matrix-core mode"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from rec_pangu_amd import hip
from rec_pangu_amd.models.layers import CompressedInteractionNet
from oracle import ref_ops as R
DEV = "cuda"
def run(B, H, D, units, mode):
    hip.set_matmul_precision(mode)
    g = torch.Generator().manual_seed(B + H)
    torch.manual_seed(B)
    cin = CompressedInteractionNet(H, units, output_dim=1)
    ld = (H * D + 63) // 64 * 64
    xbuf = torch.randn(B, ld, generator=g) * 0.5
    coef = torch.randn(B, 1, generator=g)
    n = len(units)
    def oracle(dt):
        rx = xbuf[:, :H * D].reshape(B, H, D).clone().to(dt).requires_grad_(True)
        rw = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in cin.named_parameters()}
        ref = R.cin(rx, [rw[f"cin_layer.layer_{i + 1}.weight"] for i in range(n)], [rw[f"cin_layer.layer_{i + 1}.bias"] for i in range(n)], rw["fc.weight"], rw["fc.bias"])
        (ref * coef.to(dt)).sum().backward()
        return ref.detach(), rx.grad, {k: v.grad for k, v in rw.items()}
    o64, gx64, gw64 = oracle(torch.float64)
    o32, gx32, gw32 = oracle(torch.float32)
    c = cin.to(DEV)
    dx = xbuf.to(DEV).requires_grad_(True)
    out = c(dx[:, :H * D].unflatten(1, (H, D)))
    (out * coef.to(DEV)).sum().backward()
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    msg = f"{units} {mode}: out dev {rel(out.detach(), o64):.1e} (f32 oracle {rel(o32, o64):.1e}) | dX0 dev {rel(dx.grad[:, :H*D].reshape(B,H,D), gx64):.1e} (f32 {rel(gx32, gx64):.1e})"
    for k, p in c.named_parameters():
        msg += f" | {k.replace('cin_layer.','')} {rel(p.grad, gw64[k]):.1e}"
    print(msg, flush=True)
for units in ([128, 128], [32, 32, 16], [128, 128, 128], [64, 64]):
    for mode in (("bf16x6", "fp32") if units == [32, 32, 16] else ("bf16x6",)):
        run(256, 26, 64, units, mode)
