"""Adam step over a flat arena of the m preset's size, back to back: python tools/adam_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K, lib as L
n = 74_000_000
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 0.01; m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
lib = L.load()
def run(off, label):
    pp, gg, mm, vv, ss = p[off:], g[off:], m[off:], v[off:], sh[off:]
    nn = n - off
    def step(i):
        rc = lib.s2t_adam_step(pp.data_ptr(), gg.data_ptr(), mm.data_ptr(), vv.data_ptr(), ss.data_ptr(), nn, None, 1e-3, 0.9, 0.98, 1e-8, 0.0, i, L.stream()); assert rc == 0
    for i in range(1, 4): step(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(4, 24): step(i)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print("adam %d params, %s: %.1f us = %.2f TB/s (30 B per parameter)" % (nn, label, us, nn * 30 / us / 1e6))
for _ in range(2):
    run(0, "16-byte accesses")
    run(1, "element-wise form (arrays off by one element: also unaligned)")
