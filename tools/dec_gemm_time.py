"""The decoder-side products (M = 2,560 token rows) with their real epilogues, queued back to back: python tools/dec_gemm_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; dt = torch.bfloat16; M = int(os.environ.get("M", 2560))
def timeit(fn, n=200, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = torch.Generator(device=dev).manual_seed(0)
def t(x, *s): return (torch.randn(*s, device=dev, generator=g) * x).to(dt)
D, F = 512, 2048
x = t(1, M, D); h = t(1, M, F); r = t(1, M, D); dy = t(1, M, D); dh = t(1, M, F)
wq = t(D ** -0.5, 3 * D, D); wo = t(D ** -0.5, D, D); w1 = t(D ** -0.5, F, D); w2 = t(F ** -0.5, D, F)
bq = torch.randn(3 * D, device=dev); bo = torch.randn(D, device=dev); b1 = torch.randn(F, device=dev)
res = []
res.append(("qkv  NT N=1536 K=512 bias", timeit(lambda: K.gemm(x, wq, bias=bq))))
res.append(("o    NT N=512  K=512 bias+res+drop", timeit(lambda: K.gemm(x, wo, bias=bo, residual=r, p_drop=0.1, seed=1))))
res.append(("fc1  NT N=2048 K=512 bias+relu+drop", timeit(lambda: K.gemm(x, w1, bias=b1, act=K.ACT_RELU, p_drop=0.1, seed=2))))
res.append(("fc2  NT N=512  K=2048 bias+res+drop", timeit(lambda: K.gemm(h, w2, bias=bo, residual=r, p_drop=0.1, seed=3))))
res.append(("dfc2 NN N=2048 K=512 relu_bwd", timeit(lambda: K.gemm(dy, w2, trans_b=True, act=K.ACT_RELU_BWD, aux=h))))
res.append(("dfc1 NN N=512  K=2048", timeit(lambda: K.gemm(dh, w1, trans_b=True))))
res.append(("do   NN N=512  K=512", timeit(lambda: K.gemm(dy, wo, trans_b=True))))
dq = t(1, M, 3 * D)
res.append(("dqkv NN N=512  K=1536 accumulate", timeit(lambda: K.gemm(dq, wq, trans_b=True, out=r, accumulate=True))))
print(" | ".join("%s %.1f" % (n, v) for n, v in res), "| sum %.1f us" % sum(v for _, v in res))
