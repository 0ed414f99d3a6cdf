"""One big NT GEMM in a loop (for rocprofv3 --pmc runs): python tools/one_gemm.py N K [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
N, Kd = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(24000, Kd, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, Kd, device="cuda", generator=g) * Kd ** -0.5).to(torch.bfloat16)
for _ in range(reps):
    K.gemm(a, w)
torch.cuda.synchronize()
