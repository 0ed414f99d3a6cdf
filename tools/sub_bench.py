"""Micro-benchmark of the HBM-bound subsampler kernels at the bench shape (B=64, T=1500, F=80, C=64): us + TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K


def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

dev = "cuda"
B, T, F, C = 64, 1500, 80, 64
T2, F2 = (T + 1) // 2, (F + 1) // 2
P = B * T2 * F2
dt = torch.bfloat16
x = torch.randn(B, T, F, device=dev)
w = torch.randn(C, 9, device=dev) * 0.3; bias = torch.zeros(C, device=dev)
y, sums, _ = K.conv1_fwd(x, w, bias, C, dt)
ybytes = y.numel() * 2
t = timeit(lambda: K.conv1_fwd(x, w, bias, C, dt))
print("conv1_fwd      %8.1f us  %5.2f TB/s (write y + read x)" % (t * 1e6, (ybytes + x.numel() * 4) / t / 1e12))
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev); nb = torch.zeros((), dtype=torch.int64, device=dev)
mean, rstd, scale, shift = K.bn_finalize(sums, gamma, beta, rm, rv, nb, float(P), True)
t = timeit(lambda: K.bn_apply(y, scale, shift))
print("bn_apply       %8.1f us  %5.2f TB/s" % (t * 1e6, 2 * ybytes / t / 1e12))
dyn = torch.randn_like(y)
t = timeit(lambda: K.chan_sums(y, C))
print("chan_sums m0   %8.1f us  %5.2f TB/s" % (t * 1e6, ybytes / t / 1e12))
t = timeit(lambda: K.chan_sums(y, C, dyn=dyn, mean=mean, rstd=rstd))
print("chan_sums m1   %8.1f us  %5.2f TB/s" % (t * 1e6, 2 * ybytes / t / 1e12))
s2 = K.chan_sums(y, C, dyn=dyn, mean=mean, rstd=rstd)
dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
t = timeit(lambda: K.bn_bwd_apply(dyn, y, mean, rstd, gamma, s2, dg, db, float(P)))
print("bn_bwd_apply   %8.1f us  %5.2f TB/s" % (t * 1e6, 3 * ybytes / t / 1e12))
dw = torch.zeros(C, 9, device=dev); dbias = torch.zeros(C, device=dev)
t = timeit(lambda: K.conv1_bwd(x, dyn, dw, dbias))
print("conv1_bwd      %8.1f us  %5.2f TB/s" % (t * 1e6, (ybytes + x.numel() * 4) / t / 1e12))
# conv2 forward: direct kernel vs the gathered GEMM (needs the engine's row maps: timed through the engine below)
w2 = torch.randn(C, C, 3, 3, device=dev) * 0.05
w2p = K.permute_conv_w(w2, torch.empty((C, 9 * C), dtype=dt, device=dev), C, C, 0)
y1n = torch.randn(B, T2, F2, C, device=dev).to(dt)
t = timeit(lambda: K.conv2_fwd(y1n, w2p, bias, B, T2, F2, C))
T4, F4 = (T2 + 1) // 2, (F2 + 1) // 2
print("conv2_fwd      %8.1f us  %5.2f TB/s (read y1n + write z2), %.0f TFLOP/s" % (t * 1e6, (y1n.numel() + B * T4 * F4 * C) * 2 / t / 1e12,
                                                                                   2.0 * B * T4 * F4 * C * 9 * C / t / 1e12))
w2q = K.permute_conv_w(w2, torch.empty((C, 9 * C), dtype=dt, device=dev), C, C, 1)
dpre2 = torch.randn(T4 * B * F4, C, device=dev).to(dt)
dy1n = torch.empty(B * T2 * F2, C, device=dev, dtype=dt)
t = timeit(lambda: K.conv2_dgrad(dpre2, w2q, dy1n, B, T2, F2, C, 0.1, 5))
print("conv2_dgrad    %8.1f us  %5.2f TB/s (read dpre + write dy1n), %.0f TFLOP/s" % (t * 1e6, (dy1n.numel() + dpre2.numel()) * 2 / t / 1e12,
                                                                                     2.0 * B * T4 * F4 * C * 9 * C / t / 1e12))
gw2 = torch.zeros(C, 9 * C, device=dev)
t = timeit(lambda: K.conv2_wgrad(dpre2, y1n.view(-1, C), gw2, B, T2, F2, C))
print("conv2_wgrad    %8.1f us  %5.2f TB/s (read dpre + y1n), %.0f TFLOP/s" % (t * 1e6, (y1n.numel() + dpre2.numel()) * 2 / t / 1e12,
      2.0 * T4 * B * F4 * C * 9 * C / t / 1e12))
gw = torch.zeros(C, 9, device=dev); gb = torch.zeros(C, device=dev)
t = timeit(lambda: K.conv1_bwd_bn(x, dyn, y, mean, rstd, gamma, s2, gw, gb, dg, db, float(P)))
print("conv1_bwd_bn   %8.1f us  %5.2f TB/s (read dyn + y + x)" % (t * 1e6, (2 * ybytes + x.numel() * 4) / t / 1e12))
