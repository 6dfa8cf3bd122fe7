import sys, torch, time
sys.path.insert(0, '.')
from rec_pangu_amd import hip
hip.lib()
def bench(M, N, K, lda, mode):
    hip.set_matmul_precision(mode)
    a = torch.randn(M, lda, device='cuda'); w = torch.randn(N, (K+3)//4*4, device='cuda')[:, :K]; b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    for _ in range(3): hip.linear_fwd(a, w, b, hip.ACT_RELU, K=K, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.linear_fwd(a, w, b, hip.ACT_RELU, K=K, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{mode:7s} M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
for shape in [(65536, 1024, 1677, 1728), (65536, 1728, 1024, 1024), (65536, 512, 1024, 1024), (65536, 1024, 512, 512), (65536, 256, 512, 512), (65536, 512, 256, 256), (65536, 512, 649, 704)]:
    for mode in ("bf16x3", "bf16x6"):
        bench(*shape, mode)
