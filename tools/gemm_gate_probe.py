"""Where should a product leave the 128 x 128 register-staged kernels for the 256-wide LDS-DMA kernel?  Times shapes with few 192 / 256-row
tiles (the l preset at 9,000 tokens, N = 1,024) both ways: python tools/gemm_gate_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, Kd in ((9000, 1024, 1024), (9000, 1024, 3072), (9000, 1024, 4096), (6000, 1024, 4096), (4500, 1024, 4096), (3000, 1024, 1024), (9000, 512, 2048),
                 (12000, 512, 512), (12000, 512, 2048), (6000, 2048, 512), (2560, 2048, 512), (2560, 512, 2048)):
    a = torch.randn(M, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
    wt = (torch.randn(Kd, N, device=dev, generator=g) * Kd ** -0.5).to(dt); b = torch.randn(N, device=dev, generator=g)
    out = torch.empty(M, N, device=dev, dtype=dt)
    res = []
    for mt in (100000, 1):          # 100000: never the 256-wide kernel; 1: always (when the shape is legal for it)
        old = K.set_option("gemm256_min_tiles", mt)
        res.append((timeit(lambda: K.gemm(a, w, bias=b, out=out)), timeit(lambda: K.gemm(a, wt, trans_b=True, out=out))))
        K.set_option("gemm256_min_tiles", old)
    t192 = (M + 191) // 192 * ((N + 255) // 256)
    fl = 2.0 * M * N * Kd
    print("M=%5d N=%4d K=%4d  tiles(192) %4d   NT 128-wide %6.1f us  256-wide %6.1f us   NN 128-wide %6.1f us  256-wide %6.1f us   (256-wide: %4.0f / %4.0f TF/s)"
          % (M, N, Kd, t192, res[0][0], res[1][0], res[0][1], res[1][1], fl / res[1][0] / 1e6, fl / res[1][1] / 1e6))
