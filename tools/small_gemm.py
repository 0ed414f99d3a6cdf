"""Decoder-side GEMM shapes (M = 2560 tokens): us per launch; run with S2T_GEMM_SMALL=<tile-count threshold> to move the 64x64 / 128x128 switch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
def timeit(fn, n=50, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
dev = "cuda"; dt = torch.bfloat16; M = 2560
tot = 0.0
for (N, Kd) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (8000, 512)]:
    a = torch.randn(M, Kd, device=dev).to(dt); w = torch.randn(N, Kd, device=dev).to(dt); bias = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev).to(dt); gw = torch.zeros(N, Kd, device=dev); gb = torch.zeros(N, device=dev)
    t1 = timeit(lambda: K.gemm(a, w, bias=bias)); t2 = timeit(lambda: K.gemm(dy, w, trans_b=True))
    from fbk_fairseq_st_amd.engine import _splitk
    sk = _splitk(N, Kd, M)
    t3 = timeit(lambda: K.linear_wgrad(dy, a, gw, gb, splitk=sk))
    print("N=%5d K=%5d  NT %6.1f us  NN %6.1f us  TN(sk=%d) %6.1f us" % (N, Kd, t1, t2, sk, t3)); tot += t1 + t2 + t3
print("sum %.1f us" % tot)
