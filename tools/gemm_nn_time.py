"""Data-gradient products (NN form, gemm256 TB = true) of an encoder layer, many iterations each: python tools/gemm_nn_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; dt = torch.bfloat16; M = 24000
def timeit(fn, n=100, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
for (N, Kd) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    w = torch.randn(N, Kd, device=dev).to(dt); dy = torch.randn(M, N, device=dev).to(dt); a = torch.randn(M, Kd, device=dev).to(dt)
    t_nn = min(timeit(lambda: K.gemm(dy, w, trans_b=True)) for _ in range(3))
    t_nt = min(timeit(lambda: K.gemm(a, w)) for _ in range(3))
    out.append("N=%d K=%d: NT %.1f NN %.1f" % (N, Kd, t_nt, t_nn))
print(" | ".join(out))
