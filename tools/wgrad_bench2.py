import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev = "cuda"; M = 24000
g = torch.Generator(device=dev).manual_seed(0)
def mk(n_out, n_in):
    dy = torch.randn(M, n_out, device=dev, generator=g).to(torch.bfloat16); x = torch.randn(M, n_in, device=dev, generator=g).to(torch.bfloat16)
    return dy, x, torch.zeros(n_out, n_in, device=dev), torch.zeros(n_out, device=dev)
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e6
for nprob, shape in [(16, (2048, 512)), (4, (2048, 512)), (8, (2048, 512)), (20, (2048, 512)), (32, (2048, 512)), (16, (512, 2048)), (64, (512, 512))]:
    items = [mk(*shape) for _ in range(nprob)]
    tiles = nprob * ((shape[0] + 255) // 256) * ((shape[1] + 255) // 256)
    fl = nprob * 2.0 * M * shape[0] * shape[1]
    us = t(lambda: K.wgrad_group(items))
    print("%2d x dW %s: %4d tiles  %8.1f us  %6.0f TF/s" % (nprob, shape, tiles, us, fl / us / 1e6))
