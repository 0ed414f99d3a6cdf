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
# CTC loss at the same shape: alpha/beta recursion (forward) and the gradient pass (backward)
pred, pmax, lse = K.ctc_argmax(x, want_lse=True)
Lt = 40
tg = torch.randint(4, V - 1, (B, Lt), device="cuda"); tl = torch.full((B,), Lt, device="cuda", dtype=torch.int64)
il = torch.full((B,), T, device="cuda", dtype=torch.int32)
loss, ws, _ = K.ctc_loss(x, tg, tl, il, V - 1, defer_grad=True, lse=lse)
up = torch.ones(1, device="cuda")
t = timeit(lambda: K.ctc_loss(x, tg, tl, il, V - 1, defer_grad=True, lse=lse))
print("ctc alpha/beta          %7.1f us" % t)
t = timeit(lambda: K.ctc_loss_grad(ws, up))
print("ctc_grad               %7.1f us  %5.2f TB/s (read logits + write gradient)" % (t, 2 * T * B * V * 2 / t / 1e6))
