"""Decoder-side (M = 2560 token) products: time against the reduction length K, to separate the fixed cost of a launch (prologue,
epilogue) from the cost per K-step.  python tools/small_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev = "cuda"
M = int(os.environ.get("M", 2560))


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


g = torch.Generator(device=dev).manual_seed(0)
for N in (512, 1536, 2048):
    for Kd in (64, 128, 256, 512, 1024, 2048):
        a = torch.randn(M, Kd, device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn(N, Kd, device=dev, generator=g).to(torch.bfloat16)
        wt = torch.randn(Kd, N, device=dev, generator=g).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_nt = timeit(lambda: K.gemm(a, w, bias=bias, out=out))
        t_nn = timeit(lambda: K.gemm(a, wt, trans_b=True, out=out))
        fl = 2.0 * M * N * Kd
        print("M=%d N=%4d K=%4d   NT %6.1f us %6.1f TF/s    NN %6.1f us %6.1f TF/s" % (M, N, Kd, t_nt, fl / t_nt / 1e6, t_nn, fl / t_nn / 1e6))
t0 = timeit(lambda: K.add_inplace(out, out))
print("for scale: an elementwise launch over the [%d, 2048] output takes %.1f us" % (M, t0))
