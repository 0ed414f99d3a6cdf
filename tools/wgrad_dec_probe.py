"""The decoder's grouped weight-gradient launch taken apart: the six K/V projections of the cross-attention (23,936 source tokens), the
decoder-token products (2,560 tokens) and both together, each timed alone.  python tools/wgrad_dec_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev, dt = "cuda", torch.bfloat16
TS, TD, D, F, V = 23936, 2560, 512, 2048, 8000
g = torch.Generator(device=dev).manual_seed(0)


def rnd(m, n): return (torch.randn(m, n, device=dev, generator=g) * 0.1).to(dt)


def item(tokens, n_out, n_in):
    return (rnd(tokens, n_out), rnd(tokens, n_in), torch.zeros(n_out, n_in, device=dev), torch.zeros(n_out, device=dev))


long_items = [item(TS, 2 * D, D) for _ in range(6)]
short_items = []
for _ in range(6):
    short_items += [item(TD, 3 * D, D), item(TD, D, D), item(TD, D, D), item(TD, D, D), item(TD, F, D), item(TD, D, F)]
short_items.append(item(TD, V, D))


def timeit(items, n=20):
    for _ in range(3): K.wgrad_group(items)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): K.wgrad_group(items)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    fl = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items)
    return us, fl / us / 1e6


for name, items in (("6 K/V projections over the source tokens", long_items), ("decoder-token products", short_items),
                    ("both (the decoder's launch)", long_items + short_items), ("one K/V projection", long_items[:1]),
                    ("the output embedding alone", short_items[-1:]), ("decoder-token products of one layer", short_items[:6])):
    us, tf = timeit(items)
    print("%-45s %4d products  %8.1f us  %7.1f TF/s" % (name, len(items), us, tf))
