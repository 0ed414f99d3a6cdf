"""What the dropout mask costs the gemm256 epilogue: the forward products of an encoder layer that carry one, with and without it.
python tools/gemm_drop_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; dt = torch.bfloat16; M = 24000
def timeit(fn, n=40, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, N, Kd, res, act in (("out-proj + residual", 512, 512, True, K.ACT_NONE), ("fc1 + relu", 2048, 512, False, K.ACT_RELU),
                              ("fc2 + residual", 512, 2048, True, K.ACT_NONE)):
    a = torch.randn(M, Kd, device=dev).to(dt); w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(dt); bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    out = torch.empty(M, N, device=dev, dtype=dt)
    t = [timeit(lambda: K.gemm(a, w, bias=bias, residual=r, act=act, out=out, p_drop=p, seed=3)) for p in (0.0, 0.1, 0.0, 0.1)]
    print("%-20s N=%4d K=%4d: p=0 %.1f / %.1f us   p=0.1 %.1f / %.1f us" % (name, N, Kd, t[0], t[2], t[1], t[3]))
