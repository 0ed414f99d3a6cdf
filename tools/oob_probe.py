"""Do gemm256's stores stay inside an [M][N] output whose last row tile is ragged?  The epilogue addresses a step's rows through the
buffer instruction's SCALAR offset; whether the hardware's range check covers that term decides whether rows >= M are dropped.
    python tools/oob_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for (M, N, Kd) in [(6211, 1536, 512), (23000, 640, 1280), (24000, 2048, 512), (6211, 2048, 512)]:
    a = torch.randn(M, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
    b = torch.randn(N, device=dev, generator=g)
    for name, kw in [("bias", dict(bias=b)), ("bias+drop", dict(bias=b, p_drop=0.1, seed=3))]:
        big = torch.full((M + 600, N), 7.0, dtype=dt, device=dev)
        K.gemm(a, w, out=big[:M], **kw)
        torch.cuda.synchronize()
        touched = int((big[M:] != 7.0).sum())
        rows = (big[M:] != 7.0).any(dim=1).nonzero().flatten()
        print("NT %5d x %4d x %4d %-10s guard elements overwritten: %d (rows %s)" % (M, N, Kd, name, touched,
              "-" if touched == 0 else "%d..%d" % (M + int(rows.min()), M + int(rows.max()))))
        bad += touched
    dy = torch.randn(M, N, device=dev, generator=g).to(dt)
    big = torch.full((M + 600, Kd), 7.0, dtype=dt, device=dev)
    K.gemm(dy, w, trans_b=True, out=big[:M])
    torch.cuda.synchronize()
    touched = int((big[M:] != 7.0).sum()); bad += touched
    print("NN %5d x %4d x %4d            guard elements overwritten: %d" % (M, Kd, N, touched))
print("OOB total:", bad)
sys.exit(1 if bad else 0)
