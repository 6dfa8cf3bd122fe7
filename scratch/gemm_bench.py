"""Scratch: time rp_linear_fwd / rp_linear_wgrad shapes on the GPU box (HIP events, 20 reps)."""
import sys, torch
sys.path.insert(0, ".")
from rec_pangu_amd import hip

dev = "cuda"
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

M = 65536
for mode in sys.argv[1:] or ["bf16x6"]:
    hip.set_matmul_precision(mode)
    print("mode", mode)
    for (N, K, lda, ldw, tag) in [(64, 1677, 1696, 1677, "L1 fwd (W rows unaligned)"),
                                  (64, 1680, 1696, 1680, "L1 fwd K=1680 aligned W"),
                                  (64, 1664, 1696, 1664, "L1 fwd K=1664"),
                                  (1696, 64, 64, 64, "L1 dgrad"),
                                  (1664, 64, 64, 64, "L1 dgrad N=1664"),
                                  (64, 64, 64, 64, "hidden"),
                                  (512, 649, 672, 649, "mmoe experts"),
                                  (1024, 1677, 1696, 1677, "wide L1 fwd"),
                                  (1024, 1680, 1728, 1680, "wide L1 fwd aligned"),
                                  (512, 1024, 1024, 1024, "wide L2 fwd"),
                                  (1728, 1024, 1024, 1024, "wide L1 dgrad")]:
        a = torch.randn(M, lda, device=dev)
        w = torch.randn(N, ldw, device=dev)[:, :K] if ldw != K else torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        us = t(lambda: hip.linear_fwd(a, w, None, 0, K=K, out=out))
        gb = (M * K + M * N) * 4 / us / 1e3
        print(f"  fwd {tag:28s} M={M} N={N} K={K}: {us:8.1f} us  {gb:7.1f} GB/s  {2*M*N*K/us/1e6:7.1f} TF")
    for (N, K, ldx, tag) in [(64, 1677, 1696, "L1 wgrad"), (64, 64, 64, "hidden wgrad"), (1024, 1677, 1696, "wide wgrad")]:
        x = torch.randn(M, ldx, device=dev)
        dy = torch.randn(M, N, device=dev)
        us = t(lambda: hip.linear_wgrad(dy, x, K))
        print(f"  wgrad {tag:26s} N={N} K={K}: {us:8.1f} us  {(M*K+M*N)*4/us/1e3:7.1f} GB/s  {2*M*N*K/us/1e6:7.1f} TF")
