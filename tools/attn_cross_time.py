"""Times the decoder's encoder-attention shape (Tq = 40 over Tk = 368): python tools/attn_cross_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; B, H, Tq, Tk, d = 64, 8, 40, 368, 64; D = H * d
q = torch.randn(Tq, B, D, device=dev).to(torch.bfloat16); kv = torch.randn(Tk, B, 2 * D, device=dev).to(torch.bfloat16)
k, v = kv[:, :, :D], kv[:, :, D:]
klen = torch.full((B,), Tk, dtype=torch.int32, device=dev); klen[::3] = Tk - 9
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for p in (0.0, 0.1):
    o, lse = K.attn_fwd(q, k, v, H, klen=klen, p_drop=p, seed=1)
    do = torch.randn_like(o); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    tf = timeit(lambda: K.attn_fwd(q, k, v, H, klen=klen, p_drop=p, seed=1))
    tb = timeit(lambda: K.attn_bwd(q, k, v, o, do, lse, H, dq, dkv[:, :, :D], dkv[:, :, D:], klen=klen, p_drop=p, seed=1))
    print("p_drop=%.1f  fwd %.1f us  bwd %.1f us" % (p, tf, tb))
