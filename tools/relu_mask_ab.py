"""A/B on one box: fc1 forward with ReLU (ACT_RELU) against ReLU + 1-bit record (ACT_RELU_MASK), and the fc2 data gradient through the
stored activations (ACT_RELU_BWD) against the record (ACT_RELU_BWD_MASK).  Operands rotate over SETS buffers so that the 256 MB
infinity cache does not serve them.  python tools/relu_mask_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev, dt = "cuda", torch.bfloat16
M, N, Kd, SETS = 24000, 2048, 512, 4
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(M, Kd, device=dev, generator=g).to(dt) for _ in range(SETS)]
dys = [torch.randn(M, Kd, device=dev, generator=g).to(dt) for _ in range(SETS)]
w1 = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt); b1 = torch.randn(N, device=dev, generator=g) * 0.1
w2 = (torch.randn(Kd, N, device=dev, generator=g) * 0.05).to(dt)
outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(SETS)]
das = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(SETS)]
nb = K.relu_mask_bytes(M, N, Kd)
recs = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(SETS)]


def timeit(fn, n=40):
    for i in range(SETS): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i % SETS)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for rep in range(3):
    t_plain = timeit(lambda i: K.gemm(xs[i], w1, bias=b1, act=K.ACT_RELU, p_drop=0.1, seed=5, out=outs[i]))
    t_mask = timeit(lambda i: K.gemm(xs[i], w1, bias=b1, act=K.ACT_RELU_MASK, aux_out=recs[i], p_drop=0.1, seed=5, out=outs[i]))
    t_baux = timeit(lambda i: K.gemm(dys[i], w2, trans_b=True, act=K.ACT_RELU_BWD, aux=outs[i], alpha=1.1, out=das[i]))
    t_bmask = timeit(lambda i: K.gemm(dys[i], w2, trans_b=True, act=K.ACT_RELU_BWD_MASK, aux=recs[i], alpha=1.1, out=das[i]))
    t_bnone = timeit(lambda i: K.gemm(dys[i], w2, trans_b=True, alpha=1.1, out=das[i]))
    print("fwd relu %.1f us   relu+record %.1f us   |   bwd through activations %.1f us   through record %.1f us   no mask %.1f us"
          % (t_plain, t_mask, t_baux, t_bmask, t_bnone))
