"""Run one GEMM shape repeatedly (target for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
M, N, Kd = [int(v) for v in os.environ.get("SHAPE", "24000,2048,512").split(",")]
mode = os.environ.get("MODE", "NT")
dt = torch.bfloat16
a = torch.randn(M, Kd, device="cuda").to(dt); w = torch.randn(N, Kd, device="cuda").to(dt); bias = torch.randn(N, device="cuda")
dy = torch.randn(M, N, device="cuda").to(dt); gw = torch.zeros(N, Kd, device="cuda")
for _ in range(int(os.environ.get("REPS", "10"))):
    if mode == "NT": K.gemm(a, w, bias=bias)
    elif mode == "NN": K.gemm(dy, w, trans_b=True)
    else: K.gemm(dy, a, trans_a=True, trans_b=True, out=gw, accumulate=True, splitk=8)
torch.cuda.synchronize()
