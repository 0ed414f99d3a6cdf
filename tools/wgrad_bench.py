"""Weight-gradient products of encoder layers: one grouped launch (wgrad_group.hip) vs one split-K launch per Linear."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"
M = int(os.environ.get("M", 24000)); LAYERS = int(os.environ.get("LAYERS", 4))
g = torch.Generator(device=dev).manual_seed(0)
def mk(n_out, n_in):
    dy = torch.randn(M, n_out, device=dev, generator=g).to(torch.bfloat16); x = torch.randn(M, n_in, device=dev, generator=g).to(torch.bfloat16)
    return dy, x, torch.zeros(n_out, n_in, device=dev), torch.zeros(n_out, device=dev)
items = [mk(*s) for _ in range(LAYERS) for s in ((512, 512), (1536, 512), (2048, 512), (512, 2048))]
flops = sum(2.0 * M * a[0].shape[1] * a[1].shape[1] for a in items)
def splitk(n_out, k_in, m):
    tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
    if tiles >= 256: return 1
    sk = int(max(1, min(512 // tiles, m // 256, 32)))
    return sk - sk % 8 if sk >= 8 else sk
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
tg = t(lambda: K.wgrad_group(items))
tp = t(lambda: [K.linear_wgrad(dy, x, dw, db, splitk=splitk(dw.shape[0], dw.shape[1], M)) for dy, x, dw, db in items])
print("M=%d, %d layers (%d products, %.1f GFLOP): grouped %.1f us = %.0f TF/s ; per-Linear %.1f us = %.0f TF/s" %
      (M, LAYERS, len(items), flops / 1e9, tg * 1e6, flops / tg / 1e12, tp * 1e6, flops / tp / 1e12))
