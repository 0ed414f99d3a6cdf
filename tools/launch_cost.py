"""Back-to-back launch cost of tiny kernels on one stream (us per launch) and of an 'empty' big-grid kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
def timeit(fn, n=200, w=20):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = torch.randn(1024, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
print("tiny dropout kernel (1 workgroup): %.2f us per launch" % timeit(lambda: K.dropout(x, 0.1, 1, out=y)))
big = torch.randn(24000, 512, device="cuda").to(torch.bfloat16); bo = torch.empty_like(big)
print("dropout 24.5 MB: %.2f us" % timeit(lambda: K.dropout(big, 0.1, 1, out=bo)))
g = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda")
print("ln_fwd 24000x512: %.2f us" % timeit(lambda: K.layernorm_fwd(big, g, b)))
a = torch.randn(256, 64, device="cuda").to(torch.bfloat16); w = torch.randn(64, 64, device="cuda").to(torch.bfloat16)
print("gemm 256x64x64 (4 workgroups): %.2f us" % timeit(lambda: K.gemm(a, w)))
import time
def host_cost(fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("host time per call: dropout %.2f us, layernorm_fwd %.2f us, gemm %.2f us, linear_wgrad %.2f us, attn_fwd %.2f us" % (
    host_cost(lambda: K.dropout(x, 0.1, 1, out=y)), host_cost(lambda: K.layernorm_fwd(a, torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"))),
    host_cost(lambda: K.gemm(a, w)), host_cost(lambda: K.linear_wgrad(a, a, torch.zeros(64, 64, device="cuda"), None, 1)),
    host_cost(lambda: K.attn_fwd(a.view(4, 64, 64), a.view(4, 64, 64), a.view(4, 64, 64), 1))))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): K.gemm(a, w)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
