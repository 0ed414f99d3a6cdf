"""LayerNorm backward on the encoder's activation shape, all four output variants: python tools/ln_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, D in ((24000, 512), (2560, 512), (32000, 256), (12000, 1024)):
    x = torch.randn(M, D, device=dev).to(torch.bfloat16); dy = torch.randn_like(x); dres = torch.randn_like(x)
    g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    y, mean, rstd = K.layernorm_fwd(x, g, b)
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    mb = x.numel() * 2 / 1e6
    t = timeit(lambda: K.layernorm_fwd(x, g, b)); print("M=%d D=%d fwd            %6.1f us %5.2f TB/s" % (M, D, t, 2 * mb / t))
    for name, kw, nt in (("bwd", {}, 3), ("bwd+dres", dict(dres=dres), 4), ("bwd+dres+drop", dict(dres=dres, drop=(0.1, 5)), 5)):
        t = timeit(lambda: K.layernorm_bwd(dy, x, mean, rstd, g, dg, db, **kw))
        print("M=%d D=%d %-14s %6.1f us %5.2f TB/s" % (M, D, name, t, nt * mb / t))
