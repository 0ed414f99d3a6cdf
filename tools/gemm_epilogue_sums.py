"""md5 of every gemm256 epilogue variant's output over a few shapes and launches, on whatever library S2T_HIP_LIB names, with
s2t_set_option("gemm256_sched", <argv[1]>).  tests/test_kernels_gpu.py::test_gemm256_store_data_hazard_twins compares the product
library (schedule 0) with the `make twins` builds (schedule 1: every wave in the epilogue at the same time as its SIMD partner)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
K.set_option("gemm256_sched", int(sys.argv[1]))
dev, dt = "cuda", torch.bfloat16
for (m, N, Kd) in [(24000, 512, 512), (24000, 2048, 512), (24000, 512, 2048), (23000, 640, 1280)]:
    g = torch.Generator(device=dev).manual_seed(m + N + Kd)
    a = torch.randn(m, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
    b = torch.randn(N, device=dev, generator=g); r = torch.randn(m, N, device=dev, generator=g).to(dt)
    dy = torch.randn(m, N, device=dev, generator=g).to(dt); aux = torch.randn(m, Kd, device=dev, generator=g).to(dt)
    calls = [lambda: K.gemm(a, w, bias=b), lambda: K.gemm(a, w, bias=b, p_drop=0.25, seed=9),
             lambda: K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3), lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5),
             lambda: K.gemm(dy, w, trans_b=True), lambda: K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD, aux=aux, alpha=1.25)]
    for rep in range(4):
        for i, fn in enumerate(calls):
            o = fn()
            torch.cuda.synchronize()
            print("SUM %d %d %d %d %d %s" % (m, N, Kd, i, rep, hashlib.md5(o.view(torch.int16).cpu().numpy().tobytes()).hexdigest()))
