"""Big-GEMM routes side by side (one process, interleaved rounds): 256x256x64 LDS-DMA kernel (gemm256.hip) vs the 128x128
register-staged kernels (gemm.hip) on the encoder shapes of s2t_transformer_m (M = B * T4 tokens), random data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev = "cuda"
M = int(os.environ.get("M", 24000))
ROUNDS = int(os.environ.get("ROUNDS", 5))


def time_one(fn, n=10):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def ab(name, fn, flops):
    res = {0: [], 1: []}
    for r in (0, 1):
        K.set_option("gemm256", r); fn(); fn()
    for _ in range(ROUNDS):
        for r in (1, 0):
            K.set_option("gemm256", r)
            res[r].append(time_one(fn))
    K.set_option("gemm256", 1)
    m = {r: sorted(v)[len(v) // 2] for r, v in res.items()}
    print("%-34s  new %7.1f us %7.1f TF/s   old %7.1f us %7.1f TF/s   x%.2f" %
          (name, m[1] * 1e6, flops / m[1] / 1e12, m[0] * 1e6, flops / m[0] / 1e12, m[0] / m[1]))
    return m


tot = {0: 0.0, 1: 0.0}
g = torch.Generator(device=dev).manual_seed(0)
for (N, Kd) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    a = torch.randn(M, Kd, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    fl = 2.0 * M * N * Kd
    m = ab("NT M=%d N=%d K=%d bias" % (M, N, Kd), lambda: K.gemm(a, w, bias=bias), fl)
    for r in m: tot[r] += m[r]
    if N == 2048:
        ab("NT  ... + relu + dropout", lambda: K.gemm(a, w, bias=bias, act=K.ACT_RELU, p_drop=0.1, seed=3), fl)
    if N == 512:
        ab("NT  ... + residual + dropout", lambda: K.gemm(a, w, bias=bias, residual=res, p_drop=0.15, seed=3), fl)
    dy = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    m = ab("NN M=%d N=%d K=%d (dX)" % (M, Kd, N), lambda: K.gemm(dy, w, trans_b=True), fl)
    for r in m: tot[r] += m[r]
print("sum of the 8 products: new %.1f us, old %.1f us" % (tot[1] * 1e6, tot[0] * 1e6))
