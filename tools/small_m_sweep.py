"""Tile-shape thresholds of the bf16 GEMM route at the small-batch shapes (Cfg3 at 8 x 1500: M = 3,000 encoder rows): the whole update
with s2t_set_option "gemm_small_kt" (NN / TN products: below that many 128 x 128 tiles the 64 x 64 form) and "gemm_small_nt" swept,
and the per-product times behind it.   python tools/small_m_sweep.py"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev = "cuda"


def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def products():
    g = torch.Generator(device=dev).manual_seed(0)
    M = 3000
    for N, Kd in ((512, 1536), (512, 512), (512, 2048), (2048, 512), (1536, 512)):
        dy = torch.randn(M, Kd, device=dev, generator=g).to(torch.bfloat16)
        w_kn = torch.randn(Kd, N, device=dev, generator=g).to(torch.bfloat16)       # NN: dX = dY . W, W stored [K][N]
        w_nk = torch.randn(N, Kd, device=dev, generator=g).to(torch.bfloat16)       # NT
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        line = "M=%d N=%4d K=%4d (%3d tiles of 128x128):" % (M, N, Kd, ((M + 127) // 128) * (N // 128))
        for kt, nt in ((40, 192), (100000, 100000)):
            K.set_option("gemm_small_kt", kt); K.set_option("gemm_small_nt", nt)
            t_nn = timeit(lambda: K.gemm(dy, w_kn, trans_b=True, out=out))
            t_nt = timeit(lambda: K.gemm(dy, w_nk, out=out))
            line += "   [kt %6d nt %6d] NN %5.1f us NT %5.1f us" % (kt, nt, t_nn, t_nt)
        print(line)
    K.set_option("gemm_small_kt", 40); K.set_option("gemm_small_nt", 192)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "products":
        products()
        sys.exit(0)
    products()
