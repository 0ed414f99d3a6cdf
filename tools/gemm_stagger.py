"""'CUs out of phase' experiment on gemm256: in the twin library built with
    make -C fbk_fairseq_st_amd/csrc v NAME=stagger DEFS=-DS2T_G256_STAGGER
every other CU of an XCD starts its first tile `d` ticks (10 ns) late, so that the two halves of the chip run their store bursts and their
K-loops at different times.  A launch can only get LONGER by d unless the interleaving speeds both halves up: the figure of interest is
T(d) - T(0) against d.  Prints one table per product of the headline update (M = 24,000 tokens):
    S2T_HIP_LIB=fbk_fairseq_st_amd/libs2t_hip_stagger.so python tools/gemm_stagger.py [launches per point, default 200]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K, lib as L

DEV = "cuda"
DELAYS = [0, 50, 100, 200, 400, 800]          # ticks of 10 ns


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    L.load()
    raw = ctypes.CDLL(L.LIB_PATH)
    raw.s2t_g256_set_stagger.argtypes = [ctypes.c_longlong]
    g = torch.Generator(device=DEV); g.manual_seed(1)
    M = 24000
    prods = []
    for name, N, Kd, kw in (("qkv   N 1536 K 512 bias", 1536, 512, {}),
                            ("out   N 512 K 512 bias+res+drop", 512, 512, {"res": True, "p_drop": 0.1}),
                            ("fc1   N 2048 K 512 bias+relu+drop", 2048, 512, {"act": K.ACT_RELU, "p_drop": 0.1}),
                            ("fc2   N 512 K 2048 bias+res+drop", 512, 2048, {"res": True, "p_drop": 0.1})):
        a = torch.randn(M, Kd, device=DEV, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, Kd, device=DEV, generator=g) * Kd ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=DEV, generator=g)
        r = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16) if kw.pop("res", False) else None
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        prods.append((name, 2.0 * M * N * Kd, lambda a=a, w=w, b=b, r=r, out=out, kw=kw: K.gemm(a, w, bias=b, residual=r, out=out, seed=3, **kw)))
    for name, flops, fn in prods:
        for _ in range(20):
            fn()
        acc = {d: [] for d in DELAYS}
        for rep in range(3):
            for d in DELAYS:
                assert raw.s2t_g256_set_stagger(d) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record(); torch.cuda.synchronize()
                acc[d].append(e0.elapsed_time(e1) / n * 1e3)
        raw.s2t_g256_set_stagger(0)
        t0 = min(acc[0])
        print("%s: T(0) = %.2f us (%.0f TFLOP/s)" % (name, t0, flops / t0 / 1e6))
        for d in DELAYS[1:]:
            t = min(acc[d])
            print("    d = %5.2f us   T = %.2f us   T - T(0) = %+.2f us   (runs %s)" % (d / 100.0, t, t - t0, " ".join("%.2f" % v for v in acc[d])))


if __name__ == "__main__":
    main()
