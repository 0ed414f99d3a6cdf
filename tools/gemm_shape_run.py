"""One encoder-shaped product in a loop, for rocprofv3 --pmc / --kernel-trace runs (tools/gemm_pmc.sh):
    python tools/gemm_shape_run.py <case> [reps]
cases: qkv (NT 24000x1536x512 bias) | out (NT 512x512 bias+res+drop) | fc1 (NT 2048x512 bias+relu record+drop) |
       fc2 (NT 512x2048 bias+res+drop) | dqkv (NN 512x1536) | dout (NN 512x512) | dfc1 (NN 512x2048) | dfc2 (NN 2048x512 relu record)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

case = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M = int(os.environ.get("M", 24000))
g = torch.Generator(device="cuda").manual_seed(0)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda", generator=g) * scale).to(torch.bfloat16)


SH = {"qkv": (1536, 512), "out": (512, 512), "fc1": (2048, 512), "fc2": (512, 2048),
      "dqkv": (512, 1536), "dout": (512, 512), "dfc1": (512, 2048), "dfc2": (2048, 512)}
N, Kd = SH[case]
if case[0] != "d":
    a, w = rnd(M, Kd), rnd(N, Kd, scale=Kd ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    res = rnd(M, N)
    if case == "qkv":
        fn = lambda: K.gemm(a, w, bias=bias)
    elif case == "fc1":
        nb = K.relu_mask_bytes(M, N, Kd)
        rec = torch.empty(nb, dtype=torch.uint8, device="cuda")
        fn = lambda: K.gemm(a, w, bias=bias, act=K.ACT_RELU_MASK, aux_out=rec, p_drop=0.15, seed=3)
    else:
        fn = lambda: K.gemm(a, w, bias=bias, residual=res, p_drop=0.15, seed=3)
else:
    dy, w = rnd(M, Kd), rnd(Kd, N, scale=Kd ** -0.5)       # dX[M][N] = dY[M][K] . W[K][N]
    if case == "dfc2":
        nb = K.relu_mask_bytes(M, N, Kd)
        rec = torch.randint(0, 255, (nb,), dtype=torch.uint8, device="cuda", generator=g)
        fn = lambda: K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD_MASK, aux=rec, alpha=1.0 / 0.85)
    else:
        fn = lambda: K.gemm(dy, w, trans_b=True)
K.set_option("gemm256_sched", int(os.environ.get("GEMM256_SCHED", 0)))
for _ in range(reps):
    fn()
torch.cuda.synchronize()
