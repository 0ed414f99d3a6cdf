"""Kernel micro-benchmarks on the shapes of s2t_transformer_m (B*T4 = 24000 tokens): TFLOP/s per kernel."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev = "cuda"
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

M = int(os.environ.get("M", 24000))
for dtype in ((torch.bfloat16,) if os.environ.get("BF16_ONLY") else (torch.bfloat16, torch.float32)):
    print("== dtype", dtype)
    for (N, Kd) in [(512, 512), (1536, 512), (2048, 512), (512, 2048), (512, 1280), (5001, 512), (8000, 512)]:
        mm = M if N not in (8000,) else 2560
        a = torch.randn(mm, Kd, device=dev).to(dtype); w = torch.randn(N, Kd, device=dev).to(dtype)
        bias = torch.randn(N, device=dev)
        t = timeit(lambda: K.gemm(a, w, bias=bias))
        print("NT  M=%6d N=%5d K=%5d  %8.1f us  %7.1f TF/s" % (mm, N, Kd, t * 1e6, 2 * mm * N * Kd / t / 1e12))
        dy = K.alloc_rows((mm,), N, dtype, dev); dy.copy_(torch.randn(mm, N, device=dev))     # padded rows as in the engine
        t = timeit(lambda: K.gemm(dy, w, trans_b=True))
        print("NN  M=%6d N=%5d K=%5d  %8.1f us  %7.1f TF/s" % (mm, Kd, N, t * 1e6, 2 * mm * N * Kd / t / 1e12))
        gw = torch.zeros(N, Kd, device=dev)
        for sk in (1, 4, 8):
            t = timeit(lambda: K.gemm(dy, a, trans_a=True, trans_b=True, out=gw, accumulate=True, splitk=sk))
            print("TN  M=%6d N=%5d K=%5d sk=%d %8.1f us  %7.1f TF/s" % (N, Kd, mm, sk, t * 1e6, 2 * mm * N * Kd / t / 1e12))
    B, H, T, d = M // 375, 8, 375, 64
    qkv = torch.randn(T, B, 3 * H * d, device=dev).to(dtype)
    D = H * d
    q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    t = timeit(lambda: K.attn_fwd(q, k, v, H))
    fl = 4 * B * H * T * T * d
    print("attn fwd B=%d H=%d T=%d  %8.1f us  %7.1f TF/s" % (B, H, T, t * 1e6, fl / t / 1e12))
    o, lse = K.attn_fwd(q, k, v, H)
    do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
    t = timeit(lambda: K.attn_bwd(q, k, v, o, do, lse, H, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:]))
    print("attn bwd                 %8.1f us  %7.1f TF/s (10*BHT^2d)" % (t * 1e6, 2.5 * fl / t / 1e12))
    t = timeit(lambda: K.attn_fwd(q, k, v, H, p_drop=0.1, seed=1))
    print("attn fwd + dropout       %8.1f us" % (t * 1e6))
    t = timeit(lambda: K.attn_bwd(q, k, v, o, do, lse, H, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], p_drop=0.1, seed=1))
    print("attn bwd + dropout       %8.1f us" % (t * 1e6))
    x = torch.randn(M, 512, device=dev).to(dtype); g = torch.ones(512, device=dev); b = torch.zeros(512, device=dev)
    t = timeit(lambda: K.layernorm_fwd(x, g, b))
    print("layernorm fwd M=%d D=512 %8.1f us  %6.2f TB/s" % (M, t * 1e6, 2 * x.numel() * x.element_size() / t / 1e12))
    dy = torch.randn(M, 512, device=dev).to(dtype); mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
    dg = torch.zeros(512, device=dev); db = torch.zeros(512, device=dev)
    t = timeit(lambda: K.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dy))
    print("layernorm bwd            %8.1f us  %6.2f TB/s" % (t * 1e6, 4 * x.numel() * x.element_size() / t / 1e12))
    for N in (512, 2048):
        xx = torch.randn(M, N, device=dev).to(dtype); o = torch.zeros(N, device=dev)
        t = timeit(lambda: K.colsum(xx, o))
        print("colsum N=%d              %8.1f us  %6.2f TB/s" % (N, t * 1e6, xx.numel() * xx.element_size() / t / 1e12))
    w5 = torch.randn(512, 512, device=dev).to(dtype)
    t = timeit(lambda: K.gemm(x, w5, bias=g, residual=x, p_drop=0.15, seed=3))
    print("NT 512x512 + residual + dropout epilogue  %8.1f us" % (t * 1e6))
