"""Is the 256-wide route bit-identical to the 128-wide route (same MFMA instruction, same K order, same epilogue arithmetic)?
python tools/gemm_route_equal.py -> per shape and epilogue: equal / max abs difference"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
SHAPES = [(24000, 512, 512), (24000, 1536, 512), (24000, 512, 2048), (6211, 1536, 512), (12000, 1024, 1024)]
dev, dt = "cuda", torch.bfloat16
for (M, N, Kd) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N + Kd)
    a = torch.randn(M, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
    b = torch.randn(N, device=dev, generator=g); r = torch.randn(M, N, device=dev, generator=g).to(dt)
    dy = torch.randn(M, N, device=dev, generator=g).to(dt); aux = torch.randn(M, Kd, device=dev, generator=g).to(dt)
    calls = {"bias": lambda: K.gemm(a, w, bias=b), "bias+drop": lambda: K.gemm(a, w, bias=b, p_drop=0.25, seed=9),
             "bias+res+drop": lambda: K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3),
             "relu+drop": lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5),
             "relu+res": lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, residual=r), "nn": lambda: K.gemm(dy, w, trans_b=True),
             "nn relu_bwd": lambda: K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD, aux=aux, alpha=1.25)}
    for name, fn in calls.items():
        K.set_option("gemm256", 1); x = fn()
        K.set_option("gemm256", 0); y = fn()
        K.set_option("gemm256", 1)
        torch.cuda.synchronize()
        d = (x.float() - y.float()).abs().max().item()
        print("%-28s %-14s %s" % ((M, N, Kd), name, "EQUAL" if torch.equal(x, y) else "max abs diff %.3e" % d))
