import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; dt = torch.bfloat16
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 24000
for (N, Kd) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (8000, 512)]:
    mm = M if N != 8000 else 2560
    a = torch.randn(mm, Kd, device=dev).to(dt); dy = torch.randn(mm, N, device=dev).to(dt); gw = torch.zeros(N, Kd, device=dev)
    out = []
    for sk in (1, 2, 3, 4, 6, 8, 12, 16, 32):
        t = timeit(lambda: K.gemm(dy, a, trans_a=True, trans_b=True, out=gw, accumulate=True, splitk=sk))
        out.append("sk%d %.0f" % (sk, t))
    print("dW %dx%d (tokens %d): " % (N, Kd, mm) + " | ".join(out))
