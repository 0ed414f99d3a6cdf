"""gemm256: the two K-loop schedules side by side in ONE process (interleaved rounds; s2t_set_option "gemm256_sched" 0 = two wave groups
alternating phases of 32 MFMAs, 1 = all waves software-pipelined at 16-MFMA stages).  First a bitwise check of every epilogue variant
(the schedules must give identical results: same MFMA order over K), then the encoder's eight product shapes + their epilogues.
    python tools/gemm_sched_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev, dt = "cuda", torch.bfloat16
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 7
M = int(os.environ.get("M", 24000))


def check():
    bad = 0
    for (m, N, Kd) in [(24000, 512, 512), (24000, 1536, 512), (24000, 2048, 512), (24000, 512, 2048), (6211, 1536, 512), (23000, 640, 1280), (12000, 1024, 1024)]:
        g = torch.Generator(device=dev).manual_seed(m + N + Kd)
        a = torch.randn(m, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
        b = torch.randn(N, device=dev, generator=g); r = torch.randn(m, N, device=dev, generator=g).to(dt)
        dy = torch.randn(m, N, device=dev, generator=g).to(dt); aux = torch.randn(m, Kd, device=dev, generator=g).to(dt)
        calls = [lambda: K.gemm(a, w, bias=b), lambda: K.gemm(a, w, bias=b, p_drop=0.25, seed=9),
                 lambda: K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3), lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5),
                 lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, residual=r), lambda: K.gemm(dy, w, trans_b=True),
                 lambda: K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD, aux=aux, alpha=1.25)]
        K.set_option("gemm256_sched", 0)
        want = [fn() for fn in calls]
        K.set_option("gemm256_sched", 1)
        for rep in range(6):
            for i, fn in enumerate(calls):
                got = fn()
                if not torch.equal(got, want[i]):
                    bad += 1
                    d = (got != want[i])
                    idx = d.nonzero()[:6].tolist()
                    print("MISMATCH shape %s epilogue %d launch %d: %d values differ, first at %s" % ((m, N, Kd), i, rep, int(d.sum()), idx))
    K.set_option("gemm256_sched", 0)
    print("bitwise check: %s" % ("ok" if not bad else "%d mismatching launches" % bad))
    return bad


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
        K.set_option("gemm256_sched", r); fn(); fn()
    for _ in range(ROUNDS):
        for r in (1, 0):
            K.set_option("gemm256_sched", r)
            res[r].append(time_one(fn))
    K.set_option("gemm256_sched", 0)
    m = {r: sorted(v)[len(v) // 2] for r, v in res.items()}
    print("%-40s sched1 %7.1f us %7.1f TF/s   sched0 %7.1f us %7.1f TF/s   x%.3f" %
          (name, m[1] * 1e6, flops / m[1] / 1e12, m[0] * 1e6, flops / m[0] / 1e12, m[0] / m[1]))
    return m


bad = check()
tot = {0: 0.0, 1: 0.0}
g = torch.Generator(device=dev).manual_seed(0)
for (N, Kd) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    a = torch.randn(M, Kd, device=dev, generator=g).to(dt)
    w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g).to(dt)
    fl = 2.0 * M * N * Kd
    if N == 1536:
        m = ab("NT %dx%dx%d bias (qkv)" % (M, N, Kd), lambda: K.gemm(a, w, bias=bias), fl)
    elif N == 2048:
        nb = K.relu_mask_bytes(M, N, Kd); rec = torch.empty(nb, dtype=torch.uint8, device=dev)
        m = ab("NT %dx%dx%d relu record + drop (fc1)" % (M, N, Kd), lambda: K.gemm(a, w, bias=bias, act=K.ACT_RELU_MASK, aux_out=rec, p_drop=0.15, seed=3), fl)
    else:
        m = ab("NT %dx%dx%d bias+res+drop" % (M, N, Kd), lambda: K.gemm(a, w, bias=bias, residual=res, p_drop=0.15, seed=3), fl)
    for r in m: tot[r] += m[r]
    dy = torch.randn(M, N, device=dev, generator=g).to(dt)
    if N == 2048:
        rec2 = torch.randint(0, 255, (K.relu_mask_bytes(M, N, Kd),), dtype=torch.uint8, device=dev)
        w2 = (torch.randn(Kd, N, device=dev, generator=g) * 0.05).to(dt); dy2 = torch.randn(M, Kd, device=dev, generator=g).to(dt)
        m = ab("NN %dx%dx%d relu record (dfc2)" % (M, N, Kd), lambda: K.gemm(dy2, w2, trans_b=True, act=K.ACT_RELU_BWD_MASK, aux=rec2, alpha=1 / 0.85), fl)
    else:
        m = ab("NN %dx%dx%d (dX)" % (M, Kd, N), lambda: K.gemm(dy, w, trans_b=True), fl)
    for r in m: tot[r] += m[r]
print("sum of the 8 products: sched1 %.1f us, sched0 %.1f us" % (tot[1] * 1e6, tot[0] * 1e6))
sys.exit(1 if bad else 0)
