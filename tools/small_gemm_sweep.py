"""Decoder-sized products (M = 2,560 / 3,000 / 320 rows): time against K and N, back to back, to separate the fixed cost of a
launch from the cost of a K-tile:  python tools/small_gemm_sweep.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; dt = torch.bfloat16
def timeit(fn, n=300, w=20):
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
Ms = [int(a) for a in sys.argv[1:]] or [2560]
for M in Ms:
    print("== M = %d" % M)
    for N in (512, 1536, 2048):
        row_nt, row_nn = [], []
        for Kd in (128, 512, 1024, 2048):
            x = t(1, M, Kd); w = t(Kd ** -0.5, N, Kd); wt = t(Kd ** -0.5, Kd, N)
            row_nt.append("K=%d %.1f" % (Kd, timeit(lambda: K.gemm(x, w))))
            row_nn.append("K=%d %.1f" % (Kd, timeit(lambda: K.gemm(x, wt, trans_b=True))))
        print("NT N=%-5d" % N, " | ".join(row_nt))
        print("NN N=%-5d" % N, " | ".join(row_nn))
# an empty-ish kernel for the launch floor: LayerNorm of 64 rows
x = t(1, 64, 512); gam = torch.ones(512, device=dev); bet = torch.zeros(512, device=dev)
print("launch floor (layernorm of 64 x 512): %.1f us" % timeit(lambda: K.layernorm_fwd(x, gam, bet)))
