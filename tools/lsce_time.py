"""Label-smoothed cross entropy with fused gradient at the decoder's shape: python tools/lsce_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for rows, V in ((2560, 8000), (2560, 8001), (2560, 16000)):
    x = K.alloc_rows((rows,), V, torch.bfloat16, "cuda"); x.copy_(torch.randn(rows, V, device="cuda") * 2)
    y = torch.randint(4, V, (rows,), device="cuda")
    t = timeit(lambda: K.lsce(x, y, 0.1, 1))
    print("lsce %d x %d: %.1f us  (%.2f TB/s read + write)" % (rows, V, t, 2 * rows * V * 2 / t / 1e6))
