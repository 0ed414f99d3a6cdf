"""The one-kernel attention backward (s2t_set_option "attn_bwd_fused", csrc/attention.hip: attn_bwd_fused_kernel) against the two-kernel
path it replaces and against an f32 torch reference, on the encoder's shapes; then both timed.
    python tools/attn_bwd_fused_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

dev = "cuda"


def run(B, H, T, klens, p_drop, fused, q, k, v, do, seed=7):
    K.set_option("attn_bwd_fused", 1 if fused else 0)
    kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device=dev)
    o, lse = K.attn_fwd(q, k, v, H, klen=kl, p_drop=p_drop, seed=seed)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, o, do, lse, H, dq, dk, dv, klen=kl, p_drop=p_drop, seed=seed)
    torch.cuda.synchronize()
    return o, dq, dk, dv


def reference(B, H, T, klens, q, k, v, do):
    D = q.shape[-1]; d = D // H
    qf, kf, vf = (x.float().detach().requires_grad_(True) for x in (q, k, v))
    def heads(x): return x.view(T, B, H, d).permute(1, 2, 0, 3)
    s = heads(qf) @ heads(kf).transpose(-1, -2) * d ** -0.5
    if klens is not None:
        mask = torch.arange(T, device=dev)[None, :] >= torch.tensor(klens, device=dev)[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, -1) @ heads(vf)).permute(2, 0, 1, 3).reshape(T, B, D)
    o.backward(do.float())
    return o, qf.grad, kf.grad, vf.grad


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp(min=1e-9))


def main():
    g = torch.Generator(device=dev).manual_seed(3)
    for (B, H, T, klens) in ((4, 8, 375, None), (3, 8, 375, [375, 250, 131]), (2, 4, 200, [200, 129]), (2, 8, 384, None), (2, 8, 130, None)):
        D = 64 * H
        qkv = (torch.randn(T, B, 3 * D, device=dev, generator=g) * 0.7).to(torch.bfloat16)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        do = torch.randn(T, B, D, device=dev, generator=g).to(torch.bfloat16)
        ro, rq, rk, rv = reference(B, H, T, klens, q, k, v, do)
        for p_drop in (0.0, 0.1):
            a = run(B, H, T, klens, p_drop, False, q, k, v, do)
            f = run(B, H, T, klens, p_drop, True, q, k, v, do)
            line = "B %d H %d T %d klen %s p %.1f:" % (B, H, T, klens, p_drop)
            for name, x, y in zip(("dq", "dk", "dv"), a[1:], f[1:]):
                line += "  %s fused-vs-two %.2e" % (name, rel(x, y))
            if p_drop == 0.0:
                for name, x, y in zip(("dq", "dk", "dv"), f[1:], (rq, rk, rv)):
                    line += "  %s fused-vs-f32 %.2e" % (name, rel(x, y))
                for name, x, y in zip(("dq",), a[1:2], (rq,)):
                    line += "  (two-vs-f32 %.2e)" % rel(x, y)
            print(line)
    # timing on the bench shape
    B, H, T = 64, 8, 375
    D = 64 * H
    qkv = (torch.randn(T, B, 3 * D, device=dev, generator=g) * 0.7).to(torch.bfloat16)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    do = torch.randn(T, B, D, device=dev, generator=g).to(torch.bfloat16)
    o, lse = K.attn_fwd(q, k, v, H, p_drop=0.1, seed=5)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for fused in (0, 1, 0, 1):
        K.set_option("attn_bwd_fused", fused)
        for _ in range(5):
            K.attn_bwd(q, k, v, o, do, lse, H, dq, dk, dv, p_drop=0.1, seed=5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            K.attn_bwd(q, k, v, o, do, lse, H, dq, dk, dv, p_drop=0.1, seed=5)
        e1.record(); torch.cuda.synchronize()
        print("64 x 8 x 375, p 0.1: fused %d  %.1f us per backward" % (fused, e0.elapsed_time(e1) / 50 * 1e3))
    K.set_option("attn_bwd_fused", 0)


if __name__ == "__main__":
    main()
