"""s2t_set_option "gemm_deep": the 64 x 64 bf16 GEMM form with DEPTH k-tiles requested before the first MFMA (csrc/gemm.hip) against the
two-sets-in-flight loop it replaces -- bit for bit -- and both timed, on the small products of the model (decoder-side rows, small batches).
    python tools/gemm_deep_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev = "cuda"


def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    bad = 0
    for M in (320, 640, 2560, 3000, 4000):
        for N, Kd in ((512, 512), (1536, 512), (2048, 512), (512, 2048)):
            a = torch.randn(M, Kd, device=dev, generator=g).to(torch.bfloat16)
            w_nk = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(torch.bfloat16)
            w_kn = (torch.randn(Kd, N, device=dev, generator=g) * Kd ** -0.5).to(torch.bfloat16)
            bias = torch.randn(N, device=dev, generator=g)
            res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
            outs, times = {}, {}
            for deep in (0, 1):
                K.set_option("gemm_deep", deep)
                nt = lambda: K.gemm(a, w_nk, bias=bias, residual=res, p_drop=0.1, seed=3)
                nn = lambda: K.gemm(a, w_kn, trans_b=True)
                outs[deep] = (nt().clone(), nn().clone())
                times[deep] = (timeit(nt), timeit(nn))
            same = all(torch.equal(x, y) for x, y in zip(outs[0], outs[1]))
            bad += 0 if same else 1
            print("M %4d N %4d K %4d  %s   NT %5.1f -> %5.1f us   NN %5.1f -> %5.1f us" % (M, N, Kd, "identical" if same else "DIFFERENT",
                                                                                         times[0][0], times[1][0], times[0][1], times[1][1]))
    K.set_option("gemm_deep", 1)
    print("differing shapes:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
