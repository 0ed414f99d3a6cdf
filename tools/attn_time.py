"""Times the encoder-shaped attention kernels with and without dropout: python tools/attn_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; B, H, T, d = 64, 8, 375, 64; D = H * d
qkv = torch.randn(T, B, 3 * D, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for p in (0.0, 0.1):
    o, lse = K.attn_fwd(q, k, v, H, p_drop=p, seed=1)
    do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
    tf = timeit(lambda: K.attn_fwd(q, k, v, H, p_drop=p, seed=1))
    tb = timeit(lambda: K.attn_bwd(q, k, v, o, do, lse, H, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], p_drop=p, seed=1))
    print("p_drop=%.1f  fwd %.1f us  bwd %.1f us" % (p, tf, tb))
