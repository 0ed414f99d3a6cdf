"""ARCHIVED with tools/lab/gemm4w.hip (round 5: the "gemm4w" option no longer exists in the product library).
gemm4w (four waves of 128 x 128) against gemm256 (eight of 128 x 64): every epilogue variant bit for bit, and the time of each:
    python tools/gemm4w_check.py [M]          (default 24000 and 6211)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev, dt = "cuda", torch.bfloat16


def timeit(fn, n=30, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(M, N, Kd, reps=3):
    g = torch.Generator(device=dev).manual_seed(M + N + Kd)
    a = torch.randn(M, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
    b = torch.randn(N, device=dev, generator=g); r = torch.randn(M, N, device=dev, generator=g).to(dt)
    dy = torch.randn(M, N, device=dev, generator=g).to(dt); aux = torch.randn(M, Kd, device=dev, generator=g).to(dt)
    nb = K.relu_mask_bytes(M, N, Kd)

    def mask_pair():
        rec = torch.zeros(nb, dtype=torch.uint8, device=dev)
        y = K.gemm(a, w, bias=b, act=K.ACT_RELU_MASK, aux_out=rec, p_drop=0.15, seed=3)
        dx = K.gemm(aux, wT, trans_b=True, act=K.ACT_RELU_BWD_MASK, aux=rec, alpha=1.0 / 0.85)     # an [M, N] product again: same record
        return torch.cat([y.reshape(-1), dx.reshape(-1)])
    wT = (torch.randn(Kd, N, device=dev, generator=g) * Kd ** -0.5).to(dt)       # [K][N]: dX[M][N] = aux[M][K] . wT[K][N]
    pre = torch.empty(M, N, device=dev, dtype=dt)

    def gelu_pair():
        y = K.gemm(a, w, bias=b, act=K.ACT_GELU, aux_out=pre, p_drop=0.1, seed=4)
        return torch.cat([y.reshape(-1), pre.reshape(-1)])
    acc0 = torch.randn(M, N, device=dev, generator=g).to(dt)

    def accum():
        o = acc0.clone()
        K.gemm(aux, wT, trans_b=True, out=o, accumulate=True)
        return o
    calls = [("bias", lambda: K.gemm(a, w, bias=b)), ("bias+drop", lambda: K.gemm(a, w, bias=b, p_drop=0.25, seed=9)),
             ("bias+res+drop", lambda: K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3)),
             ("bias+relu+drop", lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5)),
             ("gelu+pre+drop", gelu_pair),
             ("NN", lambda: K.gemm(aux, wT, trans_b=True)),
             ("NN relu_bwd", lambda: K.gemm(aux, wT, trans_b=True, act=K.ACT_RELU_BWD, aux=r, alpha=1.25)),
             ("NN gelu_bwd", lambda: K.gemm(aux, wT, trans_b=True, act=K.ACT_GELU_BWD, aux=r)),
             ("NN accumulate", accum)]
    if nb:
        calls.append(("relu record pair", mask_pair))
    K.set_option("gemm4w", 0)
    want = [fn() for _, fn in calls]
    t0 = [timeit(fn, 10) for _, fn in calls]
    K.set_option("gemm4w", 1)
    bad = 0
    for rep in range(reps):
        for (name, fn), wv in zip(calls, want):
            got = fn()
            nd = int((got != wv).sum())
            if nd:
                bad += 1
                d = (got.float() - wv.float()).abs()
                print("  DIFF %-16s launch %d: %d of %d values differ (max %.4g; first at %s)" % (name, rep, nd, got.numel(), float(d.max()),
                      [int(x) for x in torch.nonzero(got != wv)[0]]))
    t1 = [timeit(fn, 10) for _, fn in calls]
    K.set_option("gemm4w", 0)
    print("%d x %d x %d: %s | " % (M, N, Kd, "all variants bit-identical" if not bad else "%d DIFFERENT" % bad)
          + " ".join("%s %.1f->%.1f" % (n, x, y) for (n, _), x, y in zip(calls, t0, t1)) + " | sum %.0f -> %.0f us" % (sum(t0), sum(t1)))
    return bad


if __name__ == "__main__":
    Ms = [int(sys.argv[1])] if len(sys.argv) > 1 else [24000, 6211]
    bad = 0
    for M in Ms:
        for N, Kd in ((512, 512), (1536, 512), (2048, 512), (512, 2048), (640, 1280)):
            bad += run(M, N, Kd)
    sys.exit(1 if bad else 0)
