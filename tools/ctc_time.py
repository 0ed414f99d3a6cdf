"""CTC head passes at the bench shape (24,000 rows x 5,001 units): python tools/ctc_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
T, B, V = 375, 64, 5001
x = K.alloc_rows((T, B), V, torch.bfloat16, "cuda")
x.copy_(torch.randn(T, B, V, device="cuda") * 2)
t = timeit(lambda: K.ctc_argmax(x, want_lse=True))
print("ctc_argmax (+ row lse) %7.1f us  %5.2f TB/s" % (t, T * B * V * 2 / t / 1e6))
