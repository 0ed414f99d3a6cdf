"""A/B of GEMM variants in separate processes is noisy; this runs the shapes of one m-preset layer and prints us."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; dt = torch.bfloat16
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 24000
tot = 0
for (N, Kd) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    a = torch.randn(M, Kd, device=dev).to(dt); w = torch.randn(N, Kd, device=dev).to(dt); bias = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev).to(dt); gw = torch.zeros(N, Kd, device=dev)
    t1 = timeit(lambda: K.gemm(a, w, bias=bias)); t2 = timeit(lambda: K.gemm(dy, w, trans_b=True))
    sk = max(1, min(512 // (((N + 127) // 128) * ((Kd + 127) // 128)), M // 256, 32))
    t3 = timeit(lambda: K.gemm(dy, a, trans_a=True, trans_b=True, out=gw, accumulate=True, splitk=sk))
    f = 2 * M * N * Kd / 1e6
    print("N=%5d K=%5d  NT %7.1f us %6.0f TF | NN %7.1f us %6.0f TF | TN(sk=%d) %7.1f us %6.0f TF" % (N, Kd, t1, f / t1, t2, f / t2, sk, t3, f / t3))
    tot += t1 + t2 + t3
print("layer GEMM total %.1f us" % tot)
