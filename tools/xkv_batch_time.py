"""The six cross-attention K|V projections of the decoder as one product (N = 6 x 1,024) against six, forward and data gradient:
python tools/xkv_batch_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; dt = torch.bfloat16; M = int(os.environ.get("M", 23872)); D = 512; L = 6
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best
g = torch.Generator(device=dev).manual_seed(0)
def t(x, *s): return (torch.randn(*s, device=dev, generator=g) * x).to(dt)
enc = t(1, M, D)
ws = [t(D ** -0.5, 2 * D, D) for _ in range(L)]; bs = [torch.randn(2 * D, device=dev) for _ in range(L)]
wall = torch.cat(ws, 0); ball = torch.cat(bs, 0)
outs = [torch.empty(M, 2 * D, device=dev, dtype=dt) for _ in range(L)]; oall = torch.empty(M, L * 2 * D, device=dev, dtype=dt)
def six():
    for l in range(L): K.gemm(enc, ws[l], bias=bs[l], out=outs[l])
def one(): K.gemm(enc, wall, bias=ball, out=oall)
def cat(): torch.cat(ws, 0, out=wall); torch.cat(bs, 0, out=ball)
dkv = [t(1, M, 2 * D) for _ in range(L)]; dall = torch.cat(dkv, 1).contiguous()
denc = torch.empty(M, D, device=dev, dtype=dt)
def six_b():
    for l in range(L): K.gemm(dkv[l], ws[l], trans_b=True, out=denc, accumulate=(l > 0))
def one_b(): K.gemm(dall, wall, trans_b=True, out=denc)
a, b, c, d, e = timeit(six), timeit(one), timeit(cat), timeit(six_b), timeit(one_b)
print("forward: six products %.1f us, one product %.1f us (+ %.1f us to gather the weights) | data gradient: six %.1f us, one %.1f us" % (a, b, c, d, e))
six(); one(); torch.cuda.synchronize()
print("forward equal:", all(torch.equal(oall[:, l * 2 * D:(l + 1) * 2 * D], outs[l]) for l in range(L)))
six_b(); r6 = denc.float().clone(); one_b(); torch.cuda.synchronize()
print("data gradient: max |one - six| = %.4g on values up to %.4g" % ((denc.float() - r6).abs().max().item(), r6.abs().max().item()))
