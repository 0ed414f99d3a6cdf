"""Timing of the encoder's product shapes on whatever library S2T_HIP_LIB names (the -DS2T_X twins of gemm256: make -C fbk_fairseq_st_amd/csrc x X=n).
python tools/gemm_x_time.py <sched>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
K.set_option("gemm256_sched", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
dev, dt, M = "cuda", torch.bfloat16, 24000
g = torch.Generator(device=dev).manual_seed(0)
out = []


def timeit(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(7):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10 * 1e3)
    return sorted(ts)[3]


tot = 0.0
for (N, Kd) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    a = torch.randn(M, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
    bias = torch.randn(N, device=dev, generator=g); res = torch.randn(M, N, device=dev, generator=g).to(dt)
    dy = torch.randn(M, N, device=dev, generator=g).to(dt)
    if N == 1536:
        f = lambda: K.gemm(a, w, bias=bias)
    elif N == 2048:
        rec = torch.empty(K.relu_mask_bytes(M, N, Kd), dtype=torch.uint8, device=dev)
        f = lambda: K.gemm(a, w, bias=bias, act=K.ACT_RELU_MASK, aux_out=rec, p_drop=0.15, seed=3)
    else:
        f = lambda: K.gemm(a, w, bias=bias, residual=res, p_drop=0.15, seed=3)
    t1 = timeit(f)
    if N == 2048:
        rec2 = torch.randint(0, 255, (K.relu_mask_bytes(M, N, Kd),), dtype=torch.uint8, device=dev)
        w2 = (torch.randn(Kd, N, device=dev, generator=g) * 0.05).to(dt); dy2 = torch.randn(M, Kd, device=dev, generator=g).to(dt)
        fb = lambda: K.gemm(dy2, w2, trans_b=True, act=K.ACT_RELU_BWD_MASK, aux=rec2, alpha=1 / 0.85)
    else:
        fb = lambda: K.gemm(dy, w, trans_b=True)
    t2 = timeit(fb)
    tot += t1 + t2
    out.append("%dx%d: %.1f / %.1f" % (N, Kd, t1, t2))
print(os.environ.get("S2T_HIP_LIB", "tree lib").split("/")[-1], "| fwd / dX us |", " | ".join(out), "| sum %.1f us" % tot)
