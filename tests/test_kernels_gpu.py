"""GPU parity of every HIP kernel (through the C ABI) against the CPU oracle / plain torch fp32 CPU math.
fp32 path: 1e-4 (north_star); bf16 path: tolerance stated per test (bf16 has 8 bits of mantissa).
Integer outputs (CTC compression) are compared bit-exactly."""
import math

import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden
from oracle import int_ref, s2t_ref

pytestmark = pytest.mark.gpu

K = None


@pytest.fixture(scope="module", autouse=True)
def _k():
    global K
    from fbk_fairseq_st_amd import kernels
    K = kernels
    K._lib()
    yield


DEV = "cuda"


def rel_err(a, b):
    a = a.detach().float().cpu().double(); b = b.detach().float().cpu().double()
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 2e-2


DTYPES = [torch.float32, torch.bfloat16]


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (300, 200, 136), (77, 100, 72), (1000, 512, 256), (64, 64, 1280), (5, 3, 8)])
def test_gemm_nt_epilogues(dtype, M, N, K_):
    a = rnd(M, K_, dtype=dtype, seed=1); w = rnd(N, K_, dtype=dtype, seed=2, scale=K_ ** -0.5)
    bias = rnd(N, seed=3); res = rnd(M, N, dtype=dtype, seed=4)
    ref = a.float() @ w.float().t() + bias
    out = K.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV))
    assert rel_err(out, ref) < tol(dtype)
    out = K.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=K.ACT_RELU, residual=res.to(DEV))
    assert rel_err(out, F.relu(ref) + res.float()) < tol(dtype)
    pre = torch.empty(M, N, dtype=dtype, device=DEV)
    out = K.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=K.ACT_GELU, aux_out=pre)
    assert rel_err(out, F.gelu(ref)) < tol(dtype)
    assert rel_err(pre, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_unaligned_k(dtype):
    # K = 100 (V_tgt of the fixtures): rows are not 16-byte aligned -> guarded element loads
    a = rnd(50, 100, dtype=dtype, seed=1); w = rnd(64, 100, dtype=dtype, seed=2, scale=0.1)
    assert rel_err(K.gemm(a.to(DEV), w.to(DEV)), a.float() @ w.float().t()) < tol(dtype)
    wt = rnd(100, 64, dtype=dtype, seed=3, scale=0.1)            # NN: dX = dY[50,100] @ W[100,64]
    assert rel_err(K.gemm(a.to(DEV), wt.to(DEV), trans_b=True), a.float() @ wt.float()) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K_", [(256, 128, 192), (130, 70, 100), (2000, 256, 768)])
def test_gemm_nn_and_bwd_act(dtype, M, N, K_):
    dy = rnd(M, K_, dtype=dtype, seed=1); w = rnd(K_, N, dtype=dtype, seed=2, scale=K_ ** -0.5)
    ref = dy.float() @ w.float()
    assert rel_err(K.gemm(dy.to(DEV), w.to(DEV), trans_b=True), ref) < tol(dtype)
    aux = rnd(M, N, dtype=dtype, seed=5)
    out = K.gemm(dy.to(DEV), w.to(DEV), trans_b=True, act=K.ACT_RELU_BWD, aux=aux.to(DEV))
    assert rel_err(out, ref * (aux.float() > 0)) < tol(dtype)
    out = K.gemm(dy.to(DEV), w.to(DEV), trans_b=True, act=K.ACT_GELU_BWD, aux=aux.to(DEV))
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert rel_err(out, ref * x.grad) < tol(dtype)
    acc = rnd(M, N, dtype=dtype, seed=6)
    out = K.gemm(dy.to(DEV), w.to(DEV), trans_b=True, out=acc.clone().to(DEV), accumulate=True)
    assert rel_err(out, ref + acc.float()) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K_,split", [(128, 128, 512, 1), (256, 64, 3000, 4), (100, 72, 999, 3), (512, 2048, 700, 2)])
def test_gemm_tn_weight_grad(dtype, M, N, K_, split):
    # dW[M=N_out, N=K_in] = dY^T[K_, M] X[K_, N], f32 output, split-K with atomics, accumulate into grads
    dy = rnd(K_, M, dtype=dtype, seed=1); x = rnd(K_, N, dtype=dtype, seed=2)
    ref = dy.float().t() @ x.float()
    g0 = rnd(M, N, seed=3)
    out = K.gemm(dy.to(DEV), x.to(DEV), trans_a=True, trans_b=True, out=g0.clone().to(DEV), accumulate=True,
                 splitk=split, out_dtype=torch.float32)
    err = float((out.cpu().double() - (ref + g0).double()).abs().max() / max(1.0, float(ref.abs().max())))
    assert err < (1e-4 if dtype == torch.float32 else 1e-2)


def test_gemm_gather_scatter():
    dtype = torch.float32
    src = rnd(40, 16, dtype=dtype, seed=1)
    M, taps = 24, 3
    g = torch.Generator().manual_seed(5)
    mp = torch.randint(-1, 40, (taps, M), generator=g, dtype=torch.int32)
    w = rnd(8, taps * 16, dtype=dtype, seed=2)
    A = torch.zeros(M, taps * 16)
    for t in range(taps):
        for r in range(M):
            if mp[t, r] >= 0:
                A[r, t * 16:(t + 1) * 16] = src[mp[t, r]]
    ref = A @ w.t()
    out = K.gemm(src.to(DEV), w.to(DEV), M=M, K=taps * 16, map_a=mp.to(DEV), period_a=16)
    assert rel_err(out, ref) < 1e-4
    perm = torch.randperm(M, generator=g).to(torch.int32)
    out = K.gemm(src.to(DEV), w.to(DEV), M=M, K=taps * 16, map_a=mp.to(DEV), period_a=16, map_c=perm.to(DEV))
    ref2 = torch.zeros_like(ref); ref2[perm.long()] = ref
    assert rel_err(out, ref2) < 1e-4
    # gathered B rows for a weight-gradient style product: C[M2,N2] = Adir^T[Kp,M2] . Bsrc[mapB[k]]
    Kp = 50
    a2 = rnd(Kp, 12, seed=7); mb = torch.randint(-1, 40, (Kp,), generator=g, dtype=torch.int32)
    Bg = torch.zeros(Kp, 16)
    for k in range(Kp):
        if mb[k] >= 0:
            Bg[k] = src[mb[k]]
    out = K.gemm(a2.to(DEV), src.to(DEV), trans_a=True, trans_b=True, K=Kp, map_b=mb.to(DEV))
    assert rel_err(out, a2.t() @ Bg) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,ld", [(1234, 200, 200), (5000, 64, 64), (3001, 12, 16), (777, 8, 8), (50, 1000, 1000), (300, 3, 5)])
def test_colsum(dtype, M, N, ld):
    """wide, narrow (several rows per wave-instruction), strided-view and unaligned column counts"""
    x = rnd(M, ld, dtype=dtype, seed=1)
    out = torch.ones(N, device=DEV)
    K.colsum(x.to(DEV)[:, :N], out)
    assert rel_err(out, x.float()[:, :N].sum(0) + 1) < (1e-4 if dtype == torch.float32 else 1e-3)


# ------------------------------------------------------------------ attention
def attn_ref(q, k, v, heads, klen, causal):
    Tq, B, D = q.shape; Tk = k.shape[0]; d = D // heads
    qh = q.view(Tq, B * heads, d).transpose(0, 1) * d ** -0.5
    kh = k.reshape(Tk, B * heads, d).transpose(0, 1); vh = v.reshape(Tk, B * heads, d).transpose(0, 1)
    s = torch.bmm(qh, kh.transpose(1, 2))
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf")), 1)
    if klen is not None:
        m = torch.arange(Tk)[None, :] >= klen[:, None]
        s = s.view(B, heads, Tq, Tk).masked_fill(m[:, None, None, :], float("-inf")).view(B * heads, Tq, Tk)
    p = torch.softmax(s, -1)
    return torch.bmm(p, vh).transpose(0, 1).reshape(Tq, B, D)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("heads,d", [(2, 64), (4, 32)])
@pytest.mark.parametrize("Tq,Tk,causal,ragged", [(70, 70, False, True), (16, 16, True, True), (130, 130, True, False),
                                                  (9, 200, False, True), (375, 375, False, False), (200, 333, False, True),
                                                  (257, 129, False, True), (300, 300, True, True)])
def test_attention_fwd_bwd(dtype, heads, d, Tq, Tk, causal, ragged):
    B, D = 3, heads * d
    qkv = rnd(max(Tq, Tk), B, 3 * D, dtype=dtype, seed=1)
    q, k, v = qkv[:Tq, :, :D], qkv[:Tk, :, D:2 * D], qkv[:Tk, :, 2 * D:]
    klen = torch.tensor([Tk, max(1, Tk - 5), max(1, Tk // 2)], dtype=torch.int32) if ragged else None
    qf, kf, vf = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    ref = attn_ref(qf, kf, vf, heads, klen, causal)
    do = rnd(Tq, B, D, dtype=dtype, seed=2)
    ref.backward(do.float())
    qkv_d = qkv.to(DEV)
    qd, kd, vd = qkv_d[:Tq, :, :D], qkv_d[:Tk, :, D:2 * D], qkv_d[:Tk, :, 2 * D:]
    kl = klen.to(DEV) if klen is not None else None
    out, lse = K.attn_fwd(qd, kd, vd, heads, klen=kl, causal=causal)
    assert rel_err(out, ref) < tol(dtype)
    dqkv = torch.zeros_like(qkv_d)
    K.attn_bwd(qd, kd, vd, out, do.to(DEV), lse, heads, dqkv[:Tq, :, :D], dqkv[:Tk, :, D:2 * D], dqkv[:Tk, :, 2 * D:],
               klen=kl, causal=causal)
    t = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(dqkv[:Tq, :, :D], qf.grad) < t
    assert rel_err(dqkv[:Tk, :, D:2 * D], kf.grad) < t
    assert rel_err(dqkv[:Tk, :, 2 * D:], vf.grad) < t


def test_attention_dropout_consistency():
    """With dropout the forward/backward masks must agree: finite-difference-free check via linearity in V
    (O is linear in V for a fixed mask: O(V1+V2) = O(V1)+O(V2)) and dV = (D*P)^T dO."""
    heads, d, T, B = 2, 64, 50, 2
    D = heads * d
    q, k = rnd(T, B, D, seed=1).to(DEV), rnd(T, B, D, seed=2).to(DEV)
    v1, v2 = rnd(T, B, D, seed=3).to(DEV), rnd(T, B, D, seed=4).to(DEV)
    o1, _ = K.attn_fwd(q, k, v1, heads, p_drop=0.3, seed=77)
    o2, _ = K.attn_fwd(q, k, v2, heads, p_drop=0.3, seed=77)
    o12, lse = K.attn_fwd(q, k, v1 + v2, heads, p_drop=0.3, seed=77)
    assert rel_err(o12, o1 + o2) < 1e-4
    o0, _ = K.attn_fwd(q, k, v1, heads)
    assert rel_err(o0, o1) > 1e-2          # the mask does something
    do = rnd(T, B, D, seed=5).to(DEV)
    dq, dk, dv = [torch.empty_like(q) for _ in range(3)]
    K.attn_bwd(q, k, v1, o1, do, lse, heads, dq, dk, dv, p_drop=0.3, seed=77)
    # <dO, O(V2)> == <dV, V2> for the same mask
    lhs = float((do.double() * o2.double()).sum()); rhs = float((dv.double() * v2.double()).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_attention_dropout_consistency_long_bf16():
    """Tq >= 128 in bf16 takes the second-generation forward (attn_fwd2_kernel): its dropout mask must be the one the
    backward kernels rebuild -- <dO, O(V2)> == <dV, V2> only holds if forward and backward drop the same (query, key) pairs."""
    heads, d, T, B = 2, 64, 203, 2
    D = heads * d
    bf = torch.bfloat16
    q, k = rnd(T, B, D, dtype=bf, seed=1).to(DEV), rnd(T, B, D, dtype=bf, seed=2).to(DEV)
    v1, v2 = rnd(T, B, D, dtype=bf, seed=3).to(DEV), rnd(T, B, D, dtype=bf, seed=4).to(DEV)
    klen = torch.tensor([T, T - 37], dtype=torch.int32, device=DEV)
    o1, lse = K.attn_fwd(q, k, v1, heads, klen=klen, p_drop=0.3, seed=77)
    o2, _ = K.attn_fwd(q, k, v2, heads, klen=klen, p_drop=0.3, seed=77)
    o0, _ = K.attn_fwd(q, k, v1, heads, klen=klen)
    assert rel_err(o0, o1) > 1e-2
    do = rnd(T, B, D, dtype=bf, seed=5).to(DEV)
    dq, dk, dv = [torch.empty_like(q) for _ in range(3)]
    K.attn_bwd(q, k, v1, o1, do, lse, heads, dq, dk, dv, klen=klen, p_drop=0.3, seed=77)
    lhs = float((do.double() * o2.double()).sum()); rhs = float((dv.double() * v2.double()).sum())
    assert abs(lhs - rhs) < 2e-2 * max(1.0, abs(lhs))
    # expectation: the kept fraction is 1 - p (mask statistics), via O(V = ones) = sum of kept P / (1-p)
    ones = torch.ones_like(v1)
    oo, _ = K.attn_fwd(q, k, ones, heads, klen=klen, p_drop=0.3, seed=77)
    assert abs(float(oo.float().mean()) - 1.0) < 0.05


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,D", [(7, 64), (1000, 256), (333, 512), (50, 1024), (20, 100)])
def test_layernorm(dtype, M, D):
    x = rnd(M, D, dtype=dtype, seed=1, scale=2.0) + 0.5
    g, b = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    xf = x.float().requires_grad_(True); gf = g.clone().requires_grad_(True); bf = b.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (D,), gf, bf, 1e-5)
    dy = rnd(M, D, dtype=dtype, seed=4); dres = rnd(M, D, dtype=dtype, seed=5)
    ref.backward(dy.float())
    y, mean, rstd = K.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV))
    assert rel_err(y, ref) < tol(dtype)
    dg = torch.zeros(D, device=DEV); db = torch.zeros(D, device=DEV)
    dx = K.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, g.to(DEV), dg, db, dres=dres.to(DEV))
    assert rel_err(dx, xf.grad + dres.float()) < tol(dtype)
    assert rel_err(dg, gf.grad) < (2e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(db, bf.grad) < (2e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,D", [(333, 512), (1000, 256), (50, 1024), (21, 100)])
def test_layernorm_bwd_second_output_is_the_consumers_dropout(dtype, M, D):
    """dx_drop of s2t_layernorm_bwd == s2t_dropout(dx, p, seed), bit for bit (same element index, same rounding)"""
    x = rnd(M, D, dtype=dtype, seed=1, scale=2.0).to(DEV)
    g = (1 + 0.1 * rnd(D, seed=2)).to(DEV); b = (0.1 * rnd(D, seed=3)).to(DEV)
    dy = rnd(M, D, dtype=dtype, seed=4).to(DEV); dres = rnd(M, D, dtype=dtype, seed=5).to(DEV)
    _, mean, rstd = K.layernorm_fwd(x, g, b)
    dg = torch.zeros(D, device=DEV); db = torch.zeros(D, device=DEV)
    dx0 = K.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dres)
    dg2 = torch.zeros(D, device=DEV); db2 = torch.zeros(D, device=DEV)
    dx1, dxd = K.layernorm_bwd(dy, x, mean, rstd, g, dg2, db2, dres=dres, drop=(0.3, 4242))
    assert torch.equal(dx0, dx1)
    assert torch.equal(dxd, K.dropout(dx0, 0.3, 4242))
    kept = float((dxd != 0).float().mean())
    assert abs(kept - 0.7) < 0.03


# ------------------------------------------------------------------ conv1 + BatchNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T", [(2, 37), (3, 200)])
def test_conv1_bn(dtype, B, T):
    C, Fq = 64, 80
    x = rnd(B, T, Fq, seed=1); w = rnd(C, 1, 3, 3, seed=2, scale=0.5); bias = 0.1 * rnd(C, seed=3)
    gamma, beta = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    rm, rv = 0.1 * rnd(C, seed=6), 0.5 + torch.rand(C)
    wf = w.clone().requires_grad_(True); bf = bias.clone().requires_grad_(True)
    gf = gamma.clone().requires_grad_(True); btf = beta.clone().requires_grad_(True)
    y = F.relu(F.conv2d(x.unsqueeze(1), wf, bf, stride=2, padding=1))
    yn, nrm, nrv = s2t_ref.batch_norm2d(y, gf, btf, rm, rv, True, 0.1, 1e-5)
    dyn = rnd(*yn.shape, seed=7)
    yn.backward(dyn)
    yd, sums, _ = K.conv1_fwd(x.to(DEV), w.to(DEV), bias.to(DEV), C, dtype)
    assert rel_err(yd.permute(0, 3, 1, 2), y) < tol(dtype)
    rmd, rvd = rm.clone().to(DEV), rv.clone().to(DEV)
    nb = torch.zeros(1, dtype=torch.int64, device=DEV)
    cnt = yd.numel() // C
    mean, rstd, scale, shift = K.bn_finalize(sums, gamma.to(DEV), beta.to(DEV), rmd, rvd, nb, cnt, True)
    assert rel_err(rmd, nrm) < 1e-3 and rel_err(rvd, nrv) < 1e-3 and int(nb) == 1
    ynd = K.bn_apply(yd, scale, shift)
    assert rel_err(ynd.permute(0, 3, 1, 2), yn) < (2e-4 if dtype == torch.float32 else 3e-2)
    dynd = dyn.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
    s2 = K.chan_sums(yd, C, dyn=dynd, mean=mean, rstd=rstd)
    dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
    dpre = K.bn_bwd_apply(dynd, yd, mean, rstd, gamma.to(DEV), s2, dg, db, cnt)
    t = 5e-4 if dtype == torch.float32 else 5e-2
    assert rel_err(dg, gf.grad) < t and rel_err(db, btf.grad) < t
    dw = torch.zeros(C, 9, device=DEV); dbias = torch.zeros(C, device=DEV)
    K.conv1_bwd(x.to(DEV), dpre, dw, dbias)
    assert rel_err(dw.view(C, 1, 3, 3), wf.grad) < t
    assert rel_err(dbias, bf.grad) < t
    # the one-pass form (dpre never written): same four gradients
    dg2 = torch.zeros(C, device=DEV); db2 = torch.zeros(C, device=DEV)
    dw2 = torch.zeros(C, 9, device=DEV); dbias2 = torch.zeros(C, device=DEV)
    K.conv1_bwd_bn(x.to(DEV), dynd, yd, mean, rstd, gamma.to(DEV), s2, dw2, dbias2, dg2, db2, cnt)
    assert rel_err(dw2.view(C, 1, 3, 3), wf.grad) < t and rel_err(dbias2, bf.grad) < t
    assert rel_err(dg2, gf.grad) < t and rel_err(db2, btf.grad) < t


# ------------------------------------------------------------------ CTC compression (integers bit-exact)
def test_ctc_compress_golden_bit_exact():
    g = load_golden("ctc_compress")
    x = torch.from_numpy(g["x"]).to(DEV)          # logits (ctc_fc = identity in the fixture)
    lens = torch.from_numpy(g["lens"]).to(DEV)
    T, B, D = x.shape
    pred, pmax = K.ctc_argmax(x)
    for b in range(B):
        Lb = int(g["lens"][b])
        assert np.array_equal(pred[b, :Lb].cpu().numpy().astype(np.int64), g["pred"][b, :Lb])
    for si, strat in enumerate(("avg", "weighted", "softmax")):
        seg, rs, rl, new_len, w = K.ctc_rle(pred, pmax, lens, si)
        assert np.array_equal(new_len.cpu().numpy(), g[strat + "_new_len"])            # bit-exact
        c = int_ref.ctc_rle_c(pred.cpu().numpy(), g["lens"])
        assert np.array_equal(seg.cpu().numpy(), c["seg_id"])
        nl = c["new_len"]
        for b in range(B):
            assert np.array_equal(rs[b, :nl[b]].cpu().numpy(), c["run_start"][b, :nl[b]])
            assert np.array_equal(rl[b, :nl[b]].cpu().numpy(), c["run_len"][b, :nl[b]])
        Tout = int(new_len.max())
        out = K.ctc_compress_fwd(x, w, rs, rl, new_len, Tout)
        assert rel_err(out, torch.from_numpy(g[strat + "_out"])) < 1e-4
        dout = (2 * out).contiguous()                                                   # d/dout of sum(out^2)
        dx = torch.empty_like(x)
        K.ctc_compress_bwd(dout, w, seg, dx)
        assert rel_err(dx, torch.from_numpy(g[strat + "_grad_x"])) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_ctc_argmax_random_matches_oracle(dtype):
    T, B, V = 120, 5, 5001
    x = rnd(T, B, V, dtype=dtype, seed=3, scale=3.0)
    pred, pmax = K.ctc_argmax(x.to(DEV))
    prob = torch.softmax(x.float(), -1).transpose(0, 1)
    ref = int_ref.argmax_first_np(prob.numpy())
    assert np.array_equal(pred.cpu().numpy(), ref)
    lens = torch.tensor([120, 100, 64, 65, 1])
    seg, rs, rl, new_len, w = K.ctc_rle(pred, pmax, lens.to(DEV), 0)
    runs = int_ref.ctc_rle_np(ref, lens.numpy())
    assert [len(r) for r in runs] == new_len.cpu().tolist()


def _near_tie_rows(V, mags, gaps, seed):
    """f32 logit rows whose two largest entries are `gap` ulp apart (gap 0: equal), the larger one at the LATER index in half of
    the rows and at the earlier one in the other half; everything else at least 1.5 below.  Returns x [R, V], meta rows
    (magnitude, gap, later_is_larger)."""
    rs = np.random.RandomState(seed)
    rows, meta = [], []
    for mag in mags:
        for gap in gaps:
            for later in (False, True):
                for rep in range(24):
                    x = (rs.randn(V) * 0.7 - 4.0).astype(np.float32)
                    top = np.float32(mag * (0.75 + 0.5 * rs.rand()))
                    lo = top
                    for _ in range(gap):
                        lo = np.nextafter(lo, np.float32(-np.inf), dtype=np.float32)
                    x = np.minimum(x, lo - np.float32(1.5))
                    i, j = sorted(rs.choice(V, 2, replace=False))
                    x[i], x[j] = (lo, top) if later else (top, lo)
                    rows.append(x); meta.append((mag, gap, later))
    return np.stack(rows), meta


def test_ctc_argmax_near_ties_of_a_5001_wide_row():
    """The reference takes arg-max of the f32 softmax OUTPUT (conv_transformer.py:282-284): two distinct logits whose probabilities
    round to the same float are a tie there, and the FIRST index wins.  The kernel evaluates the same formula, exp(x - max) / sum in
    f32 with first-index ties, in its own arithmetic: its sum (and exp) differ from torch's in the last bits, so whether two
    probabilities one ulp apart collapse into one float can come out differently -- on either side (the reference's own answer for
    such a row changes with the platform's exp and summation order).  Pinned here: exact logit ties give the first index; a
    disagreement is only ever between the SAME two candidates, and only where the reference's two probabilities are at most one
    ulp apart (its decision is at rounding level); how often that happens on rows built to provoke it is measured and bounded."""
    V = 5001
    x, meta = _near_tie_rows(V, mags=(0.05, 0.3, 1.5, 6.0, 20.0), gaps=(0, 1, 2, 4), seed=5)
    R = x.shape[0]
    xt = torch.from_numpy(x)
    prob = torch.softmax(xt, -1)
    ref = prob.argmax(-1).numpy()                                    # first index among equal maxima
    pred, _ = K.ctc_argmax(xt.view(R, 1, V).to(DEV))                 # [B = 1, T = R]
    got = pred.view(-1).cpu().numpy().astype(np.int64)
    top2 = torch.topk(prob, 2, dim=-1).values.numpy()
    ulps = np.abs(top2[:, 0].view(np.int32).astype(np.int64) - top2[:, 1].view(np.int32).astype(np.int64))   # distance of the reference's two probabilities
    gap = np.array([m[1] for m in meta]); mag = np.array([m[0] for m in meta])
    bad = got != ref
    assert not bad[gap == 0].any(), "exact logit ties must give the first index"
    order = np.argsort(-x, axis=1)[:, :2]
    assert all(got[r] in order[r] for r in np.nonzero(bad)[0]), "a disagreement must stay between the two near-tied candidates"
    assert not bad[ulps > 1].any(), "rows whose reference probabilities are more than one ulp apart must agree"
    near = (gap > 0) & (ulps <= 1)
    rate = float(bad[near].mean()) if near.any() else 0.0
    print("MEASURED near-tie arg-max: %d rows with unequal top logits, %d of them with reference probabilities <= 1 ulp apart, "
          "%d disagreements (%.1f %% of those; by top-logit magnitude: %s)" % (
              int((gap > 0).sum()), int(near.sum()), int(bad.sum()), 100 * rate,
              ", ".join("%g: %d/%d" % (m, int(bad[near & (mag == m)].sum()), int((near & (mag == m)).sum())) for m in sorted(set(mag)))))
    assert rate <= 0.35, rate


@pytest.mark.parametrize("V", [37, 2048, 5001, 5120, 8000, 9001])
def test_ctc_argmax_padded_rows_single_pass_and_three_pass(V):
    """rows with a padded stride (what the CTC head writes) take the single-pass kernel up to 8192 units, the three-pass kernel
    above; both must give the first arg-max of the f32 softmax, its value and the row log-sum-exp, ties included"""
    T, B = 33, 3
    x = rnd(T, B, V, dtype=torch.bfloat16, seed=V, scale=3.0)
    x[5, 1, V - 1] = x[5, 1].max() + 1.0                              # the maximum in the last column (next to the row padding)
    x[6, 2, 3] = x[6, 2, V // 2] = x[6, 2].max() + 0.5                # a tie: the first index wins
    xd = K.alloc_rows((T, B), V, torch.bfloat16, DEV)
    xd.copy_(x.to(DEV))
    pred, pmax, lse = K.ctc_argmax(xd, want_lse=True)
    prob = torch.softmax(x.float(), -1).transpose(0, 1)
    ref = int_ref.argmax_first_np(prob.numpy())
    assert np.array_equal(pred.cpu().numpy(), ref)
    assert int(pred[1, 5]) == V - 1 and int(pred[2, 6]) == 3
    want_p = torch.gather(prob, 2, torch.from_numpy(ref).long().unsqueeze(-1)).squeeze(-1)
    assert float(((pmax.cpu() - want_p).abs() / want_p).max()) <= 1e-5
    assert float((lse.cpu().view(T, B) - torch.logsumexp(x.float(), -1)).abs().max()) <= 1e-4


def test_ctc_rle_long_runs():
    # runs crossing the 64-frame chunks of the wave-parallel scan
    B, T = 4, 300
    pred = torch.zeros(B, T, dtype=torch.int32)
    pred[1] = torch.arange(T) // 70
    pred[2] = torch.arange(T) % 2
    pred[3, 150:] = 9
    lens = torch.tensor([300, 300, 299, 151])
    seg, rs, rl, new_len, w = K.ctc_rle(pred.to(DEV), torch.ones(B, T, device=DEV), lens.to(DEV), 0)
    c = int_ref.ctc_rle_c(pred.numpy(), lens.numpy())
    assert np.array_equal(new_len.cpu().numpy(), c["new_len"])
    assert np.array_equal(seg.cpu().numpy(), c["seg_id"])


# ------------------------------------------------------------------ losses
@pytest.mark.parametrize("dtype", DTYPES)
def test_ctc_loss(dtype):
    T, B, V, Lm = 60, 5, 40, 12
    blank = V - 1
    logits = rnd(T, B, V, dtype=dtype, seed=1, scale=2.0)
    g = torch.Generator().manual_seed(2)
    tgt = torch.randint(4, blank, (B, Lm), generator=g)
    tl = torch.tensor([12, 7, 1, 10, 3]); il = torch.tensor([60, 45, 30, 5, 17])
    tgt[3, :10] = 5                                   # 10 repeats need 19 frames > 5 -> infeasible -> zero_infinity
    lf = logits.float().requires_grad_(True)
    ref = s2t_ref.ctc_loss_sum(lf, tgt, il, tl, blank)
    ref.backward()
    loss, grad, nll = K.ctc_loss(logits.to(DEV), tgt.to(DEV), tl.to(DEV), il.to(torch.int32).to(DEV), blank)
    assert abs(float(loss) - float(ref)) < (1e-4 if dtype == torch.float32 else 2e-3) * abs(float(ref))
    assert rel_err(grad, lf.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    assert not math.isfinite(float(nll[3]))
    # two-call form: loss in forward, gradient in backward times the upstream device scalar
    loss2, ws, _ = K.ctc_loss(logits.to(DEV), tgt.to(DEV), tl.to(DEV), il.to(torch.int32).to(DEV), blank, defer_grad=True)
    assert abs(float(loss2) - float(loss)) <= 1e-6 * abs(float(loss))          # f32 atomics add the per-utterance terms in any order
    g1 = K.ctc_loss_grad(ws, torch.ones(1, device=DEV))
    assert torch.equal(g1, grad)
    g2 = K.ctc_loss_grad(ws, torch.full((1,), 0.25, device=DEV))
    assert rel_err(g2, 0.25 * lf.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    # row log-sum-exps as a by-product of the arg-max pass, reused by the loss
    _, _, lse = K.ctc_argmax(logits.to(DEV), want_lse=True)
    assert rel_err(lse, torch.logsumexp(logits.float(), -1).reshape(-1)) < 1e-5
    loss3, grad3, _ = K.ctc_loss(logits.to(DEV), tgt.to(DEV), tl.to(DEV), il.to(torch.int32).to(DEV), blank, lse=lse)
    assert abs(float(loss3) - float(loss)) <= 1e-6 * abs(float(loss)) and torch.equal(grad3, grad)


@pytest.mark.parametrize("Lm,T", [(20, 70), (40, 130), (100, 330), (200, 520), (400, 900), (511, 1100)])
def test_ctc_loss_transcript_lengths_of_every_lane_width(Lm, T):
    """The alpha / beta recursion keeps 1, 2, 4, 8 or 16 extended-target positions per lane (transcripts of up to 31 / 63 / 127 / 255 /
    511 units): every width against torch's float64 F.ctc_loss (the reference's call, CTC_loss.py:143-151) -- ragged frame counts,
    repeated units (no skip across equal labels), one empty and one infeasible transcript."""
    B, V = 6, 50
    blank = V - 1
    g = torch.Generator().manual_seed(Lm)
    logits = torch.randn(T, B, V, generator=g) * 2.0
    tgt = torch.randint(0, blank, (B, Lm), generator=g)
    tgt[1, : Lm // 2] = 7                                             # a long run of one unit: needs a blank between every two
    tl = torch.tensor([Lm, Lm // 2, max(Lm // 3, 1), 0, Lm, 1])
    il = torch.tensor([T, T, T - 7, T // 2, Lm // 2, 3])                # utterance 4: fewer frames than units -> infeasible
    lp = torch.log_softmax(logits.double(), -1).requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(lp, tgt, il, tl, blank=blank, reduction="sum", zero_infinity=True)
    ref.backward()
    ref = ref.detach()
    gl = lp.grad - lp.detach().exp() * lp.grad.sum(-1, keepdim=True)   # d/dlogits from d/dlog-probs
    loss, grad, nll = K.ctc_loss(logits.to(DEV), tgt.to(DEV), tl.to(DEV), il.to(torch.int32).to(DEV), blank)
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref)), (float(loss), float(ref))
    assert rel_err(grad, gl.float()) < 1e-4
    assert not math.isfinite(float(nll[4])) and math.isfinite(float(nll[3]))
    per = torch.nn.functional.ctc_loss(lp.detach(), tgt, il, tl, blank=blank, reduction="none", zero_infinity=False)
    ok = torch.isfinite(per)
    assert torch.allclose(nll.cpu()[ok].double(), per[ok], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("V", [100, 8000])
def test_lsce(dtype, V):
    rows = 37
    logits = rnd(rows, V, dtype=dtype, seed=1, scale=2.0)
    g = torch.Generator().manual_seed(2)
    tgt = torch.randint(4, V, (rows,), generator=g); tgt[5] = 1; tgt[20] = 1
    lf = logits.float().requires_grad_(True)
    loss, nll = s2t_ref.label_smoothed_nll(lf, tgt, 0.1, 1)
    loss.backward()
    sums, dl = K.lsce(logits.to(DEV), tgt.to(DEV), 0.1, 1)
    assert abs(float(sums[0]) - float(loss)) < 1e-4 * abs(float(loss))
    assert abs(float(sums[1]) - float(nll)) < 1e-4 * abs(float(nll))
    assert rel_err(dl, lf.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding(dtype):
    V, D, B, Ln, pad = 50, 64, 3, 9, 1
    W = rnd(V, D, dtype=dtype, seed=1)
    g = torch.Generator().manual_seed(2)
    tok = torch.randint(2, V, (B, Ln), generator=g); tok[1, 6:] = pad; tok[2, 3:] = pad
    table = s2t_ref.sinusoid_table(pad + 1 + Ln, D, pad)
    ref = (math.sqrt(D) * W.float()[tok] + table[s2t_ref.token_positions(tok, pad)]).transpose(0, 1)
    out = K.embed_fwd(tok.to(DEV), W.to(DEV), table.to(DEV), math.sqrt(D), pad)
    assert rel_err(out, ref) < tol(dtype)
    dout = rnd(Ln, B, D, dtype=dtype, seed=3)
    dW = torch.zeros(V, D, device=DEV)
    K.embed_bwd(tok.to(DEV), dout.to(DEV), dW, math.sqrt(D), pad)
    refg = torch.zeros(V, D)
    refg.index_put_((tok.t().reshape(-1),), math.sqrt(D) * dout.float().reshape(-1, D), accumulate=True)
    refg[pad] = 0
    assert rel_err(dW, refg) < 1e-4


def test_dropout_statistics_and_replay():
    x = torch.ones(1 << 20, device=DEV)
    y1 = K.dropout(x, 0.25, seed=5); y2 = K.dropout(x, 0.25, seed=5); y3 = K.dropout(x, 0.25, seed=6)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = float((y1 > 0).float().mean())
    assert abs(keep - 0.75) < 5e-3 and abs(float(y1.max()) - 1 / 0.75) < 1e-6


# ------------------------------------------------------------------ optimizer
def test_gradnorm_clip_adam():
    n = 100003
    p = rnd(n, seed=1); g = rnd(n, seed=2); m = torch.zeros(n); v = torch.zeros(n)
    pd, gd, md, vd = p.to(DEV), g.to(DEV), m.to(DEV), v.to(DEV)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    ws = torch.zeros(1, dtype=torch.float64, device=DEV); out2 = torch.zeros(2, device=DEV)
    scale = 1.0 / 15
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    for step in (1, 2, 3):
        K.grad_norm_clip(gd, scale, 0.5, ws, out2)
        gn, gl = s2t_ref.clip_grad_norm([g * scale], 0.5)
        assert abs(float(out2[0]) - float(gn)) < 1e-4 * float(gn)
        K.adam_step(pd, gd, md, vd, shadow, out2, 5e-4, 0.9, 0.98, 1e-8, 1e-4, step)
        pr, mr, vr = s2t_ref.adam_step(pr, gl[0], mr, vr, step, 5e-4, wd=1e-4)
    assert rel_err(pd, pr) < 1e-5 and rel_err(md, mr) < 1e-5 and rel_err(vd, vr) < 1e-5
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))


def test_adam_at_arena_size():
    """the vector loop of adam_kernel takes two 16-byte groups per lane and step once the grid is capped (> 16.8 M parameters): every
    element of a 40 M-parameter arena (a size that is not a multiple of four: the element-wise tail too) against the same arithmetic in torch"""
    n = 40_000_007
    g0 = torch.Generator(device=DEV).manual_seed(3)
    p = torch.randn(n, device=DEV, generator=g0); g = torch.randn(n, device=DEV, generator=g0) * 0.1
    m = torch.randn(n, device=DEV, generator=g0) * 0.01; v = torch.rand(n, device=DEV, generator=g0) * 1e-3
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    out2 = torch.tensor([0.0, 0.5], device=DEV)
    lr, b1, b2, eps, wd, step = 5e-4, 0.9, 0.98, 1e-8, 1e-2, 7
    gs = g * 0.5
    mr = b1 * m + (1 - b1) * gs; vr = b2 * v + (1 - b2) * gs * gs
    ss = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
    pr = (p - wd * lr * p) - ss * mr / (vr.sqrt() + eps)
    K.adam_step(p, g, m, v, shadow, out2, lr, b1, b2, eps, wd, step)
    for mine, ref in ((p, pr), (m, mr), (v, vr)):
        assert float((mine - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(shadow, p.to(torch.bfloat16))


def test_add_pos_and_permutes():
    T, B, D = 11, 3, 64
    x = rnd(T, B, D, seed=1); lens = torch.tensor([11, 7, 1])
    table = s2t_ref.sinusoid_table(T + 1, D, 0)
    ref = x + table[s2t_ref.audio_positions(lens, T)].transpose(0, 1)
    out = K.add_pos(x.clone().to(DEV), table.to(DEV), lens.to(torch.int32).to(DEV))
    assert rel_err(out, ref) < 1e-6
    # positions + dropout in one pass, out of place (both dtypes, a width that takes the element-wise path too): bit for bit
    # the separate kernels; and the backward twin: dropout + activation gradient in one pass
    for dtype, Dw in ((torch.bfloat16, 64), (torch.float32, 64), (torch.bfloat16, 20), (torch.float32, 6)):
        xs = rnd(T, B, Dw, dtype=dtype, seed=4).to(DEV); tb = s2t_ref.sinusoid_table(T + 1, Dw, 0).to(DEV); l32 = lens.to(torch.int32).to(DEV)
        two = K.dropout(K.add_pos(xs.clone(), tb, l32), 0.3, 99)
        one = K.add_pos(xs, tb, l32, out=torch.empty_like(xs), p_drop=0.3, seed=99)
        assert torch.equal(one, two) and 0.5 < float((one != 0).float().mean()) < 0.85
        dy = rnd(T * B, Dw, dtype=dtype, seed=5).to(DEV); yv = rnd(T * B, Dw, dtype=dtype, seed=6).to(DEV)
        for act in (1, 2):
            assert torch.equal(K.act_bwd(dy, yv, act, 0.3, 99), K.act_bwd(K.dropout(dy, 0.3, 99), yv, act))
    N, C, Fq = 5, 64, 20
    w = rnd(N, C * Fq, seed=2)
    wp = K.permute_cf(w.to(DEV), torch.empty(N, C * Fq, device=DEV), N, C, Fq, 0)
    assert torch.equal(wp.cpu(), w.view(N, C, Fq).transpose(1, 2).reshape(N, -1))
    acc = torch.ones(N, C * Fq, device=DEV)
    K.permute_cf(wp, acc, N, C, Fq, 1)
    assert rel_err(acc, w + 1) < 1e-6
    w2 = rnd(8, 16, 3, 3, seed=3)
    f = K.permute_conv_w(w2.to(DEV), torch.empty(8, 9 * 16, device=DEV), 8, 16, 0)
    assert torch.equal(f.cpu(), w2.permute(0, 2, 3, 1).reshape(8, -1))
    back = torch.zeros(8, 16, 3, 3, device=DEV)
    K.permute_conv_w(f, back, 8, 16, 2)
    assert torch.equal(back.cpu(), w2)


@pytest.mark.parametrize("route", [1, 0], ids=["gemm256", "gemm128"])
@pytest.mark.parametrize("M,N,K_", [(24000, 384, 192), (12000, 768, 128), (23000, 640, 1280), (36800, 256, 512), (24000, 512, 512), (6211, 1536, 512), (24000, 2048, 128)])
def test_gemm_big_products_both_routes(M, N, K_, route):
    """bf16 NT / NN products large enough for the 256 x 256 x 64 LDS-DMA kernel (gemm256.hip: ragged last row / column tiles,
    2 .. 20 K-tiles, every epilogue) against plain f32 math; route 0 sends the same calls to the 128 x 128 kernels (gemm.hip)."""
    dtype = torch.bfloat16
    old = K.set_option("gemm256", route)
    try:
        a = rnd(M, K_, dtype=dtype, seed=1); w = rnd(N, K_, dtype=dtype, seed=2, scale=K_ ** -0.5)
        bias = rnd(N, seed=3); res = rnd(M, N, dtype=dtype, seed=4)
        ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
        ref = a.float() @ w.float().t() + bias
        assert rel_err(K.gemm(ad, wd, bias=bd), ref) < 2e-2
        out = K.gemm(ad, wd, bias=bd, act=K.ACT_RELU, residual=rd)
        assert rel_err(out, F.relu(ref) + res.float()) < 2e-2
        pre = torch.empty(M, N, dtype=dtype, device=DEV)
        out = K.gemm(ad, wd, bias=bd, act=K.ACT_GELU, aux_out=pre)
        assert rel_err(out, F.gelu(ref)) < 2e-2 and rel_err(pre, ref) < 2e-2
        out32 = K.gemm(ad, wd, out_dtype=torch.float32)                     # f32 output rows stay on the gemm.hip kernels
        assert rel_err(out32, a.float() @ w.float().t()) < 2e-3
        # dropout in the epilogue: same mask as the standalone kernel on the same index space
        y = K.gemm(ad, wd, bias=bd, p_drop=0.25, seed=9)
        y0 = K.dropout(K.gemm(ad, wd, bias=bd), 0.25, 9)
        assert rel_err(y, y0) < 2e-2
        # NN: dX = act_bwd(dY . W) with W stored [K][N] (the weight as it lies in memory), accumulate into an existing buffer
        dy = rnd(M, N, dtype=dtype, seed=5); dyd = dy.to(DEV)
        refx = dy.float() @ w.float()
        assert rel_err(K.gemm(dyd, wd, trans_b=True), refx) < 2e-2
        aux = rnd(M, K_, dtype=dtype, seed=6)
        out = K.gemm(dyd, wd, trans_b=True, act=K.ACT_RELU_BWD, aux=aux.to(DEV), alpha=1.25)
        assert rel_err(out, torch.where(aux.float() > 0, 1.25 * refx, torch.zeros_like(refx))) < 2e-2
        base = rnd(M, K_, dtype=dtype, seed=7)
        acc = base.to(DEV).clone()
        K.gemm(dyd, wd, trans_b=True, out=acc, accumulate=True)
        assert rel_err(acc, refx + base.float()) < 2e-2
    finally:
        K.set_option("gemm256", old)


@pytest.mark.parametrize("M,V", [(24000, 5001), (12345, 1003)])
def test_gemm_big_logit_rows_odd_vocabulary_both_routes(M, V):
    """the CTC head at full size (conv_transformer.py:279: ctc_fc over 24,000 frames, V_src = 5,001): logit rows in a padded buffer,
    the 256-wide kernel's last 16-byte store of a row lands in the row padding; against the 128-wide route and f32 math"""
    dtype, D = torch.bfloat16, 512
    g = torch.Generator(device=DEV).manual_seed(V)
    x = torch.randn(M, D, device=DEV, generator=g).to(dtype); w = (torch.randn(V, D, device=DEV, generator=g) * D ** -0.5).to(dtype)
    b = torch.randn(V, device=DEV, generator=g)
    outs = []
    for route in (1, 0):
        old = K.set_option("gemm256", route)
        try:
            out = K.alloc_rows((M,), V, dtype, DEV, zero=True)
            K.gemm(x, w, bias=b, out=out)
            outs.append(out)
        finally:
            K.set_option("gemm256", old)
    assert rel_err(outs[0], outs[1]) < 1e-2
    ref = x[:2048].float() @ w.float().t() + b
    assert rel_err(outs[0][:2048], ref) < 2e-2
    base = outs[0].as_strided((M, K.padded_cols(V, dtype)), (K.padded_cols(V, dtype), 1))
    assert bool(torch.isfinite(base.float()).all())


@pytest.mark.parametrize("tokens", [24000, 3000, 2560, 777])
def test_wgrad_group_matches_per_linear_gradients(tokens):
    """one grouped launch for the parameter gradients of several Linears (csrc/wgrad_group.hip): dW += dY^T X, db += colsum(dY),
    vs plain f32 math on the same bf16 operands; token counts that are not a multiple of the 64-deep K-tile, output shapes that are
    not multiples of the 256 x 256 tile, row-padded dY (logit rows), accumulation into existing gradients; the tail round of a
    launch is cut along the token range and meets in f32 atomics, so reruns agree to f32 rounding, not bit for bit"""
    shapes = [(512, 512, True), (1536, 512, True), (2048, 512, True), (512, 2048, True), (1001, 512, False), (264, 1280, True)]
    items, refs = [], []
    for k, (n_out, n_in, has_b) in enumerate(shapes):
        dy = rnd(tokens, n_out, dtype=torch.bfloat16, seed=10 + k)
        x = rnd(tokens, n_in, dtype=torch.bfloat16, seed=20 + k, scale=0.5)
        dyd = K.alloc_rows((tokens,), n_out, torch.bfloat16, DEV); dyd.copy_(dy)            # row stride padded to 16 bytes
        dw0 = rnd(n_out, n_in, seed=30 + k); db0 = rnd(n_out, seed=40 + k)
        items.append((dyd, x.to(DEV), dw0.to(DEV).clone(), db0.to(DEV).clone() if has_b else None))
        refs.append((dw0.double() + dy.double().t() @ x.double(), db0.double() + dy.double().sum(0)))
    K.wgrad_group(items)
    for (dy, x, dw, db), (rw, rb) in zip(items, refs):
        assert rel_err(dw, rw) < 2e-5 * max(1.0, tokens ** 0.5 / 8), (tuple(dw.shape), rel_err(dw, rw))
        if db is not None:
            assert rel_err(db, rb) < 1e-4
    again = [(dy, x, torch.zeros_like(dw), None) for (dy, x, dw, db) in items]
    K.wgrad_group(again); first = [a[2].clone() for a in again]
    for a in again:
        a[2].zero_()
    K.wgrad_group(again)
    assert all(rel_err(a[2], f) < 1e-6 for a, f in zip(again, first))
    # a list that fills whole rounds of 256 tiles takes no atomics at all: bit-identical reruns
    big = [(items[2][0], items[2][1], torch.zeros(2048, 512, device=DEV), None) for _ in range(16)]     # 16 x 16 tiles
    K.wgrad_group(big); ref0 = big[0][2].clone()
    for b in big:
        b[2].zero_()
    K.wgrad_group(big)
    assert all(torch.equal(b[2], ref0) for b in big)


@pytest.mark.parametrize("t_long,t_short,n_long", [(23936, 2560, 6), (6001, 333, 2), (9000, 1000, 1)])
def test_wgrad_group_mixed_reduction_lengths(t_long, t_short, n_long):
    """the decoder's launch: K/V projections of the cross-attention over the source tokens (long reductions) next to the products over
    its own tokens (short ones).  The work list pours the long dWs into what the short tiles leave of every workgroup's share
    (wgrad_group.hip, layout_fill): token ranges cut at arbitrary K-tiles, pieces meeting in f32 atomics, bias gradients included"""
    g = torch.Generator(device=DEV).manual_seed(t_long)

    def mk(tokens, n_out, n_in, has_b=True):
        dy = (torch.randn(tokens, n_out, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
        x = (torch.randn(tokens, n_in, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
        return (dy, x, torch.randn(n_out, n_in, device=DEV, generator=g), torch.randn(n_out, device=DEV, generator=g) if has_b else None)

    items = [mk(t_long, 1024, 512) for _ in range(n_long)]
    for _ in range(3):
        items += [mk(t_short, 1536, 512), mk(t_short, 512, 512), mk(t_short, 2048, 512), mk(t_short, 512, 2048)]
    items.append(mk(t_short, 1000, 512, has_b=False))
    refs = [(dw.double() + dy.double().t() @ x.double(), None if db is None else db.double() + dy.double().sum(0)) for dy, x, dw, db in items]
    K.wgrad_group(items)
    for (dy, x, dw, db), (rw, rb) in zip(items, refs):
        assert rel_err(dw, rw) < 2e-5 * max(1.0, dy.shape[0] ** 0.5 / 8), (tuple(dw.shape), dy.shape[0], rel_err(dw, rw))
        if db is not None:
            assert rel_err(db, rb) < 1e-4


@pytest.mark.parametrize("tokens", [8000, 961, 30])
def test_wgrad_group_f32_matches_per_linear_gradients(tokens):
    """f32 mode's grouped weight gradients (csrc/wgrad_f32.hip): dW += dY^T X, db += colsum(dY) for a list of
    Linears against f64 math on the same f32 operands -- token counts that are not a multiple of the 32-token stage, outputs that
    are not multiples of the 128 x 128 tile (the CTC head's 5,001 rows, row-padded), accumulation into existing gradients -- and,
    every tile having one owner, bit-identical reruns (a list whose longest tiles leave a round partly filled cuts that round's tiles
    along the tokens and meets them in f32 atomics: test below)"""
    shapes = [(256, 256, True), (768, 256, True), (2048, 256, True), (256, 2048, True), (5001, 256, True), (260, 1280, False)]
    items, refs = [], []
    for k, (n_out, n_in, has_b) in enumerate(shapes):
        dy = rnd(tokens, n_out, seed=10 + k)
        x = rnd(tokens, n_in, seed=20 + k, scale=0.5)
        dyd = K.alloc_rows((tokens,), n_out, torch.float32, DEV); dyd.copy_(dy)             # row stride padded to 16 bytes
        dw0 = rnd(n_out, n_in, seed=30 + k); db0 = rnd(n_out, seed=40 + k)
        assert K.wgrad_group_ok(dyd, x.to(DEV))
        items.append((dyd, x.to(DEV), dw0.to(DEV).clone(), db0.to(DEV).clone() if has_b else None))
        refs.append((dw0.double() + dy.double().t() @ x.double(), db0.double() + dy.double().sum(0)))
    K.wgrad_group(items)
    for (dy, x, dw, db), (rw, rb) in zip(items, refs):
        assert rel_err(dw, rw) < 2e-6 * max(1.0, tokens ** 0.5 / 8), (tuple(dw.shape), rel_err(dw, rw))
        if db is not None:
            assert rel_err(db, rb) < 1e-5
    again = [(dy, x, torch.zeros_like(dw), None if db is None else torch.zeros_like(db)) for (dy, x, dw, db) in items]
    K.wgrad_group(again); first = [(a[2].clone(), None if a[3] is None else a[3].clone()) for a in again]
    for a in again:
        a[2].zero_()
        if a[3] is not None:
            a[3].zero_()
    K.wgrad_group(again)
    for a, (fw, fb) in zip(again, first):
        assert torch.equal(a[2], fw) and (fb is None or torch.equal(a[3], fb))


def test_wgrad_group_f32_cut_round():
    """1,060 equal long tiles on 512 workgroups: the 36 tiles of the third round are cut along the tokens (f32 atomics)"""
    g = torch.Generator(device=DEV).manual_seed(7)
    items = []
    for _ in range(33):
        dy = torch.randn(2048, 512, device=DEV, generator=g) * 0.5; x = torch.randn(2048, 1024, device=DEV, generator=g) * 0.5
        items.append((dy, x, torch.randn(512, 1024, device=DEV, generator=g), torch.randn(512, device=DEV, generator=g)))
    items.append((items[0][0][:, :256].contiguous(), items[0][1][:, :256].contiguous(), torch.zeros(256, 256, device=DEV), None))
    refs = [(dw.double() + dy.double().t() @ x.double(), None if db is None else db.double() + dy.double().sum(0)) for dy, x, dw, db in items]
    K.wgrad_group(items)
    for (dy, x, dw, db), (rw, rb) in zip(items, refs):
        assert rel_err(dw, rw) < 5e-6, rel_err(dw, rw)
        assert db is None or rel_err(db, rb) < 1e-5


TURN_SHAPES = [(24000, 384, 192), (24000, 512, 512), (24000, 1536, 512), (24000, 2048, 512), (24000, 512, 2048), (23000, 640, 1280),
               (6211, 1536, 512), (36800, 256, 512), (24000, 2048, 128), (12000, 1024, 1024)]


@pytest.mark.parametrize("M,N,K_", TURN_SHAPES)
def test_gemm256_epilogues_against_the_128_wide_route_and_themselves(M, N, K_):
    """tools/gemm_turn_check.py as a test: every epilogue variant of gemm256 (stores leave in a lane order turned through wave-private
    LDS slots, masked epilogue per quad, unmasked one straight-line) on ten shapes, six launches each.  The 128-wide route (gemm.hip)
    runs the same MFMA instruction over K in the same order and the same epilogue arithmetic (gemm_epilogue.hpp), so the two routes
    must agree BIT FOR BIT, and every repeated launch must reproduce the first one (a store path that races shows up as a few
    wrong values in some launches)."""
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(M + N + K_)
    a = torch.randn(M, K_, device=DEV, generator=g).to(dt); w = (torch.randn(N, K_, device=DEV, generator=g) * K_ ** -0.5).to(dt)
    b = torch.randn(N, device=DEV, generator=g); r = torch.randn(M, N, device=DEV, generator=g).to(dt)
    dy = torch.randn(M, N, device=DEV, generator=g).to(dt); aux = torch.randn(M, K_, device=DEV, generator=g).to(dt)
    calls = [lambda s: K.gemm(a, w, bias=b), lambda s: K.gemm(a, w, bias=b, p_drop=0.25, seed=9 + s),
             lambda s: K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3), lambda s: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5),
             lambda s: K.gemm(a, w, bias=b, act=K.ACT_RELU, residual=r), lambda s: K.gemm(dy, w, trans_b=True),
             lambda s: K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD, aux=aux, alpha=1.25)]
    old = K.set_option("gemm256", 0)
    try:
        want = [[fn(rep) for fn in calls] for rep in range(2)]           # the seed of the second variant changes with the launch index
        K.set_option("gemm256", 1)
        for rep in range(6):
            for i, fn in enumerate(calls):
                got = fn(rep % 2)
                assert torch.equal(got, want[rep % 2][i]), "epilogue %d, launch %d: %d of %d values differ from the 128-wide route" % (
                    i, rep, int((got != want[rep % 2][i]).sum()), got.numel())
    finally:
        K.set_option("gemm256", old)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("log_probs", [True, False])
def test_softmax_rows_forward_and_backward(dtype, log_probs):
    """s2t_log_softmax / s2t_softmax_probs / s2t_softmax_bwd (get_normalized_probs with a gradient) against float64 autograd; padded rows"""
    rows, V = 37, 1003
    x = K.alloc_rows((rows,), V, dtype, DEV)
    x.copy_((torch.randn(rows, V, generator=torch.Generator().manual_seed(3)) * 3).to(dtype))
    xr = x.detach().double().cpu().requires_grad_(True)
    ref = torch.log_softmax(xr, -1) if log_probs else torch.softmax(xr, -1)
    gout = torch.randn(rows, V, generator=torch.Generator().manual_seed(4))
    ref.backward(gout.double())
    xd = x.detach().requires_grad_(True)
    out = K.NormalizedProbs.apply(xd, log_probs)
    assert rel_err(out, ref.detach()) < 1e-5
    out.backward(gout.to(DEV))
    assert rel_err(xd.grad, xr.grad) < tol(dtype)


def test_gemm256_product_epilogues_soak_against_the_128_wide_route():
    """tools/gemm_soak.py as a test: 200 launches of every masked / operand-reading / accumulating / 1-bit-record epilogue of the
    SHIPPED gemm256 per shape, half of them beside a store-only kernel on a second stream (the CTC side stream of a training step),
    each compared bit for bit with the 128-wide route -- zero differing launches (the product route's sample had been 6).
    The 2,000-launch run of the same function is committed as profiles/r05_gemm_soak.txt"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gemm_soak
    bad, total = gemm_soak.soak(200, shapes=[(24000, 2048, 512), (24000, 512, 512)], verbose=False)
    assert total == 2 * 6 * 200 and bad == 0, "%d of %d launches differ from the 128-wide route" % (bad, total)


@pytest.mark.slow
def test_gemm256_store_data_hazard_twins():
    """Round 3 left "wrong values when the two wave groups' epilogues overlap" unexplained; this is its reproducer.  `make twins` builds
    gemm256 with a second K-loop schedule in which all eight waves -- both waves of every SIMD -- run their epilogues at the same
    time (s2t_set_option "gemm256_sched" 1; not in the product library, whose schedule keeps one epilogue per SIMD at a time):
      libs2t_hip_sched1_nohold.so   the epilogue as round 3 had it: hipcc re-uses a 16-byte store's data registers two instructions
                                    after the store (the wait states the ISA asks for).  With the SIMD partner storing too, the
                                    younger wave's stores leave with the NEXT step's f32 intermediates in their first dwords, lanes
                                    12-15 of every 16-lane row (the data beats read last; tools/gemm_sched_diff.py decodes it):
                                    thousands of wrong values in nearly every launch of a masked or operand-reading epilogue.
      libs2t_hip_sched1.so          stores from a four-deep register ring that is untouched for three steps (the product's epilogue):
                                    the raw-f32 garbage is gone; a residue of stale-but-valid values (same lanes, ~1e-5 of the
                                    elements, a few launches in a hundred) remains in the operand-reading variants, i.e. a second
                                    mechanism is still open -- which is why concurrent epilogues stay out of the product.
    Asserted: the product library is bit-exact against the 128-wide route (test above); without the hold the hazard shows; the hold
    removes at least three quarters of the differing launches.  Counts are printed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "fbk_fairseq_st_amd")

    def sums(lib, sched):
        env = dict(os.environ)
        if lib:
            env["S2T_HIP_LIB"] = os.path.join(pkg, lib)
            if not os.path.exists(env["S2T_HIP_LIB"]):
                pytest.skip("diagnostic twins not built (make -C fbk_fairseq_st_amd/csrc twins; lib.build(twins=True))")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_epilogue_sums.py"), str(sched)], env=env,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return [l for l in out.stdout.splitlines() if l.startswith("SUM ")]

    want = sums(None, 0)
    assert len(want) == 4 * 6 * 4
    raw = sums("libs2t_hip_sched1_nohold.so", 1)
    held = sums("libs2t_hip_sched1.so", 1)
    n_raw = sum(a != b for a, b in zip(want, raw))
    n_held = sum(a != b for a, b in zip(want, held))
    plain = [k for k, (a, b) in enumerate(zip(want, held)) if a != b and a.split()[4] in ("0", "4")]
    print("MEASURED store-data hazard (both waves of a SIMD in their epilogues): %d of %d launches differ from the product library "
          "without the register hold, %d with it" % (n_raw, len(want), n_held))
    assert not plain, "the plain epilogues (no mask, no operand stream) must be exact under either arrangement"
    if n_raw == 0:
        pytest.xfail("the hazard did not show on this box")
    assert n_held * 4 <= n_raw, (n_held, n_raw)


def test_gemm256_is_deterministic_under_load():
    """the LDS-DMA pipeline orders its reads by counted waits and barriers only: 40 back-to-back launches on fresh random data
    must reproduce the first result bit for bit (a read that overtakes its DMA shows up as rare wrong tiles)"""
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(24000, 512, device=DEV, generator=g).to(dtype); w = (torch.randn(2048, 512, device=DEV, generator=g) * 0.05).to(dtype)
    a2 = torch.randn(24000, 2048, device=DEV, generator=g).to(dtype); w2 = (torch.randn(512, 2048, device=DEV, generator=g) * 0.02).to(dtype)
    first, first2, firstx = K.gemm(a, w).clone(), K.gemm(a2, w2).clone(), K.gemm(a2, w, trans_b=True).clone()
    ref = (a[:4096].float() @ w.float().t())
    assert rel_err(first[:4096], ref) < 2e-2
    for _ in range(40):
        assert torch.equal(K.gemm(a, w), first)
        assert torch.equal(K.gemm(a2, w2), first2)
        assert torch.equal(K.gemm(a2, w, trans_b=True), firstx)


@pytest.mark.parametrize("M,N,K_", [(24000, 2048, 512), (23936, 2048, 512), (6211, 1536, 512), (24000, 1000, 128), (50000, 256, 192)])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_gemm_relu_one_bit_record(M, N, K_, p_drop):
    """fc1 / fc2 of the FFN (transformer_layer.py:128-136) with the ReLU decision kept as one bit per activation: the forward product
    equals the ACT_RELU one bit for bit, and the data gradient through the record equals the one that re-reads the activations"""
    dtype = torch.bfloat16
    nb = K.relu_mask_bytes(M, N, K_)
    assert nb > 0 and nb % 8192 == 0 and nb * 8 >= M * N
    g = torch.Generator(device=DEV).manual_seed(M + N)
    x = torch.randn(M, K_, device=DEV, generator=g).to(dtype); w = (torch.randn(N, K_, device=DEV, generator=g) * K_ ** -0.5).to(dtype)
    b = torch.randn(N, device=DEV, generator=g) * 0.1
    dy = torch.randn(M, K_, device=DEV, generator=g).to(dtype); w2 = (torch.randn(K_, N, device=DEV, generator=g) * 0.05).to(dtype)
    a_ref = K.gemm(x, w, bias=b, act=K.ACT_RELU, p_drop=p_drop, seed=77)
    rec = torch.full((nb,), 0xA5, dtype=torch.uint8, device=DEV)
    a = K.gemm(x, w, bias=b, act=K.ACT_RELU_MASK, aux_out=rec, p_drop=p_drop, seed=77)
    assert torch.equal(a, a_ref)
    frac = float((a > 0).float().mean())
    assert (0.4 if p_drop == 0 else 0.35) < frac < 0.55
    alpha = 1.0 / (1.0 - p_drop)
    da_ref = K.gemm(dy, w2, trans_b=True, act=K.ACT_RELU_BWD, aux=a, alpha=alpha)
    da = K.gemm(dy, w2, trans_b=True, act=K.ACT_RELU_BWD_MASK, aux=rec, alpha=alpha)
    assert torch.equal(da, da_ref)
    assert torch.equal(da != 0, (a > 0) & (K.gemm(dy, w2, trans_b=True, alpha=alpha) != 0))


def test_gemm_relu_one_bit_record_refused_where_the_kernel_does_not_run():
    """decoder-side products (M = 2,560) are not on the 256-wide kernel: no record, and the codes are refused loudly"""
    assert K.relu_mask_bytes(2560, 2048, 512) == 0 and K.relu_mask_bytes(24000, 2048, 100) == 0
    x = torch.randn(2560, 512, device=DEV).to(torch.bfloat16); w = torch.randn(2048, 512, device=DEV).to(torch.bfloat16)
    from fbk_fairseq_st_amd import lib as L
    lib = L.load()
    rec = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV); out = torch.empty(2560, 2048, device=DEV, dtype=torch.bfloat16)
    rc = lib.s2t_gemm_gather(1, 1, 0, 0, 2560, 2048, 512, x.data_ptr(), 512, w.data_ptr(), 512, out.data_ptr(), 2048, None, None, 0,
                             None, rec.data_ptr(), 0, K.ACT_RELU_MASK, 0, 1, 1.0, None, 0, None, None, 0.0, 0, L.stream())
    assert rc == -95                       # S2T_ENOTSUP


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("V", [1001, 523])
def test_gemm_ctc_head_shapes_odd_vocabulary(dtype, V):
    """The CTC head with V_src = 5001-like vocabularies: logit rows live in a padded buffer (K.alloc_rows).  The weight
    gradient reads whole 16-byte column chunks that hang over into the row padding, the data gradient has a K tail that
    is accumulated by a second launch (gemm.hip: s2t_gemm_gather)."""
    M, D = 3000, 256
    g = torch.Generator().manual_seed(11)
    base = torch.full((M, K.padded_cols(V, dtype)), float("nan"), dtype=dtype, device=DEV)      # poisoned row padding
    dl = base[:, :V]
    dl_h = (torch.randn(M, V, generator=g) * 0.1).to(dtype)
    dl.copy_(dl_h)
    x = rnd(M, D, dtype=dtype, seed=2); w = rnd(V, D, dtype=dtype, seed=3, scale=D ** -0.5)
    gw = torch.zeros(V, D, device=DEV)
    K.gemm(dl, x.to(DEV), trans_a=True, trans_b=True, out=gw, accumulate=True, splitk=4)
    ref = dl_h.float().t() @ x.float()
    assert float((gw.cpu() - ref).abs().max()) < (1e-3 if dtype == torch.float32 else 3e-2) * max(1.0, float(ref.abs().max()))
    dx = K.gemm(dl, w.to(DEV), trans_b=True)
    assert rel_err(dx, dl_h.float() @ w.float()) < tol(dtype)
    acc = rnd(M, D, dtype=dtype, seed=6)
    dx = K.gemm(dl, w.to(DEV), trans_b=True, out=acc.clone().to(DEV), accumulate=True)
    assert rel_err(dx, dl_h.float() @ w.float() + acc.float()) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_out,n_in,tokens,split", [(512, 512, 3000, 4), (1536, 512, 2048, 2), (100, 72, 999, 3), (256, 2048, 1280, 1),
                                                     (1001, 256, 3000, 4)])
def test_linear_wgrad_fused_bias_gradient(dtype, n_out, n_in, tokens, split):
    """s2t_linear_wgrad: dW += dY^T X and db += colsum(dY) in one pass (row sums ride on the MFMA A operand); shapes cover the
    fused kernel, the guarded fallback (+ separate column-sum pass) and an odd vocabulary with padded rows."""
    g = torch.Generator().manual_seed(5)
    base = torch.full((tokens, K.padded_cols(n_out, dtype)), float("nan"), dtype=dtype, device=DEV)
    dy = base[:, :n_out]
    dy_h = (torch.randn(tokens, n_out, generator=g) * 0.5).to(dtype)
    dy.copy_(dy_h)
    x = rnd(tokens, n_in, dtype=dtype, seed=2)
    gw0 = rnd(n_out, n_in, seed=3); gb0 = rnd(n_out, seed=4)
    gw = gw0.clone().to(DEV); gb = gb0.clone().to(DEV)
    K.linear_wgrad(dy, x.to(DEV), gw, gb, splitk=split)
    rw = dy_h.float().t() @ x.float() + gw0
    rb = dy_h.float().sum(0) + gb0
    t = 1e-4 if dtype == torch.float32 else 1e-2
    assert float((gw.cpu() - rw).abs().max()) < t * max(1.0, float(rw.abs().max()))
    assert float((gb.cpu() - rb).abs().max()) < t * max(1.0, float(rb.abs().max()))
    gw2 = gw0.clone().to(DEV)
    K.linear_wgrad(dy, x.to(DEV), gw2, None, splitk=split)           # no bias: weight gradient only
    assert float((gw2.cpu() - rw).abs().max()) < t * max(1.0, float(rw.abs().max()))


def test_attention_v2_head_to_xcd_mapping():
    """B*H multiple of 8: the second-generation kernels re-map workgroups so that the tiles of a head share an XCD
    (attention.hip head_xcd_remap); results must not depend on it."""
    heads, d, B, Tq, Tk = 4, 64, 4, 260, 200
    D = heads * d
    bf = torch.bfloat16
    q = rnd(Tq, B, D, dtype=bf, seed=1); k = rnd(Tk, B, D, dtype=bf, seed=2); v = rnd(Tk, B, D, dtype=bf, seed=3)
    klen = torch.tensor([Tk, Tk - 9, 150, 131], dtype=torch.int32)
    qf, kf, vf = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    ref = attn_ref(qf, kf, vf, heads, klen, False)
    do = rnd(Tq, B, D, dtype=bf, seed=4)
    ref.backward(do.float())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    out, lse = K.attn_fwd(qd, kd, vd, heads, klen=klen.to(DEV))
    assert rel_err(out, ref) < 2e-2
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    K.attn_bwd(qd, kd, vd, out, do.to(DEV), lse, heads, dq, dk, dv, klen=klen.to(DEV))
    assert rel_err(dq, qf.grad) < 3e-2 and rel_err(dk, kf.grad) < 3e-2 and rel_err(dv, vf.grad) < 3e-2
    for b in range(1, B):                                # padded keys receive exact zeros
        assert float(dk[int(klen[b]):, b].abs().max()) == 0.0 and float(dv[int(klen[b]):, b].abs().max()) == 0.0


def test_augment_kernel_reproduces_the_reference_batches():
    """G15 on the GPU: TimeStretch + SpecAugment through s2t_augment, seeded like the reference run: identical batches (exact)."""
    import random
    from helpers import load_golden
    from fbk_fairseq_st_amd.augment import SpecAugment, TimeStretch
    g = load_golden("augment")
    for ci in range(int(g["ncases"])):
        sa, ts = g["c%d_sa" % ci], g["c%d_ts" % ci]
        batch = {"net_input": {"src_tokens": torch.from_numpy(g["c%d_in" % ci]).to(DEV), "src_lengths": torch.from_numpy(g["c%d_lens" % ci])},
                 "nframes": int(g["c%d_lens" % ci].sum())}
        random.seed(100 + ci); np.random.seed(200 + ci)
        if ts[0] >= 0:
            batch = TimeStretch(float(ts[0]), int(ts[1]), float(ts[2]), float(ts[3]))(batch)
            assert torch.equal(batch["net_input"]["src_tokens"].cpu(), torch.from_numpy(g["c%d_ts_tokens" % ci]))
            assert batch["net_input"]["src_lengths"].tolist() == g["c%d_ts_lengths" % ci].tolist()
            assert batch["nframes"] == int(g["c%d_ts_lengths" % ci].sum())
        if sa[0] >= 0:
            batch = SpecAugment(int(sa[0]), int(sa[1]), int(sa[2]), int(sa[3]), float(sa[4]))(batch)
        assert torch.equal(batch["net_input"]["src_tokens"].cpu(), torch.from_numpy(g["c%d_out" % ci])), ci


@pytest.mark.parametrize("B,T2,F2", [(3, 37, 40), (2, 10, 20), (5, 64, 40), (1, 1, 40)])
def test_conv2_wgrad_all_taps_kernel(B, T2, F2):
    """s2t_conv2_wgrad (bf16, 64 channels) against torch's conv2d weight gradient in fp32: odd T2 (last output row sees one input
    row less), a single row, the 40-mel (F2 = 20) geometry."""
    C = 64
    T4, F4 = (T2 + 1) // 2, (F2 + 1) // 2
    bf = torch.bfloat16
    y1n = rnd(B, T2, F2, C, dtype=bf, seed=1)
    dpre = rnd(T4, B, F4, C, dtype=bf, seed=2, scale=0.3)
    gw = torch.zeros(C, 9 * C, device=DEV)
    assert K.conv2_wgrad(dpre.to(DEV), y1n.to(DEV).view(-1, C), gw, B, T2, F2, C)
    ref = torch.nn.grad.conv2d_weight(y1n.float().permute(0, 3, 1, 2), (C, C, 3, 3), dpre.float().permute(1, 3, 0, 2), stride=2, padding=1)
    got = gw.cpu().view(C, 9, C).permute(0, 2, 1).reshape(C, C, 3, 3)          # [co][tap][ci] -> [co][ci][kh][kw]
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-3, err
    K.conv2_wgrad(dpre.to(DEV), y1n.to(DEV).view(-1, C), gw, B, T2, F2, C)     # accumulates
    assert float((gw.cpu().view(C, 9, C).permute(0, 2, 1).reshape(C, C, 3, 3) - 2 * ref).abs().max() / ref.abs().max()) < 4e-3
    assert not K.conv2_wgrad(dpre.float().to(DEV), y1n.float().to(DEV).view(-1, C), gw, B, T2, F2, C)      # fp32: not covered


@pytest.mark.parametrize("B,T2,F2", [(3, 37, 40), (2, 10, 20), (5, 64, 40), (1, 1, 40), (2, 23, 33), (64, 50, 40)])
@pytest.mark.parametrize("act", ["relu", "gelu"])
def test_conv2_forward_direct_kernel(B, T2, F2, act):
    """s2t_conv2_fwd (bf16, 64 channels: input rows staged once in LDS) against torch's conv2d in fp32 on the same bf16 operands:
    odd T2 / F2 (zero padding on every side), a single input row, the 40-mel geometry, more units than workgroups."""
    C = 64
    T4, F4 = (T2 + 1) // 2, (F2 + 1) // 2
    bf = torch.bfloat16
    y1n = rnd(B, T2, F2, C, dtype=bf, seed=1)
    w = rnd(C, C, 3, 3, seed=2, scale=0.06)
    bias = rnd(C, seed=3, scale=0.2)
    w2p = K.permute_conv_w(w.to(DEV), torch.empty((C, 9 * C), dtype=bf, device=DEV), C, C, 0)
    A = K.ACT_GELU if act == "gelu" else K.ACT_RELU
    out = K.conv2_fwd(y1n.to(DEV), w2p, bias.to(DEV), B, T2, F2, C, A)
    assert out is not None
    z2, pre = out
    wq = w2p.float().cpu().view(C, 9, C).permute(0, 2, 1).reshape(C, C, 3, 3)     # the bf16-rounded weights the kernel multiplies
    ref_pre = torch.nn.functional.conv2d(y1n.float().permute(0, 3, 1, 2), wq, bias, stride=2, padding=1)       # [B, C, T4, F4]
    ref_pre = ref_pre.permute(2, 0, 3, 1).reshape(T4 * B * F4, C)                   # rows (t4, b, f4)
    if act == "gelu":
        assert float((pre.float().cpu() - ref_pre).abs().max()) <= 2e-2 * float(ref_pre.abs().max())
        ref = torch.nn.functional.gelu(pre.float().cpu())                          # gelu of the STORED pre-activation
    else:
        assert pre is None
        ref = torch.relu(ref_pre)
    err = float((z2.float().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err
    assert K.conv2_fwd(y1n.float().to(DEV), w2p.float(), bias.to(DEV), B, T2, F2, C, A) is None      # fp32: not covered


@pytest.mark.parametrize("B,T2,F2", [(3, 37, 40), (2, 10, 20), (5, 64, 40), (1, 1, 40), (2, 23, 33), (64, 50, 40)])
def test_conv2_data_gradient_direct_kernel(B, T2, F2):
    """s2t_conv2_dgrad against torch's conv2d input gradient in fp32 on the same bf16 operands (all four pixel-parity classes in one
    launch, odd sizes, a single row), and with dropout against s2t_dropout of the unmasked result: the same mask (the kernel scales
    the f32 sums before its one rounding, s2t_dropout the already rounded values: one bf16 ulp apart)"""
    C = 64
    T4, F4 = (T2 + 1) // 2, (F2 + 1) // 2
    bf = torch.bfloat16
    dpre = rnd(T4, B, F4, C, dtype=bf, seed=2, scale=0.3)
    w = rnd(C, C, 3, 3, seed=2, scale=0.06)
    w2q = K.permute_conv_w(w.to(DEV), torch.empty((C, 9 * C), dtype=bf, device=DEV), C, C, 1)
    dy = torch.full((B * T2 * F2, C), float("nan"), dtype=bf, device=DEV)               # every element must be written
    assert K.conv2_dgrad(dpre.to(DEV).view(-1, C), w2q, dy, B, T2, F2, C)
    wq = w.to(bf).float()                                                              # the bf16-rounded weights the kernel multiplies
    ref = torch.nn.grad.conv2d_input((B, C, T2, F2), wq, dpre.float().permute(1, 3, 0, 2), stride=2, padding=1)    # [B, C, T2, F2]
    ref = ref.permute(0, 2, 3, 1).reshape(B * T2 * F2, C)
    assert not torch.isnan(dy.float()).any()
    err = float((dy.float().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err
    dyd = torch.empty_like(dy)
    assert K.conv2_dgrad(dpre.to(DEV).view(-1, C), w2q, dyd, B, T2, F2, C, 0.1, 77)
    want = K.dropout(dy, 0.1, 77)
    keep = K.dropout(torch.ones_like(dy), 0.1, 77) != 0
    assert torch.equal(dyd != 0, keep & (dy != 0)) or float(((dyd != 0) != (keep & (dy != 0))).float().mean()) < 1e-4     # (tiny values may round to 0)
    assert float((dyd.float() - want.float()).abs().max()) <= 2 ** -7 * float(want.float().abs().max())
    assert not K.conv2_dgrad(dpre.float().to(DEV).view(-1, C), w2q.float(), dy.float(), B, T2, F2, C)     # fp32: not covered


# ------------------------------------------------------------------ ConvAttention2D pieces (csrc/attn2d.hip)
def _planes(t, B, T, Fq, C, ch):
    """[M, C] channels-last rows (t, b, f) -> [B, T, Fq] plane of channel ch"""
    return t.view(T, B, Fq, C)[..., ch].permute(1, 0, 2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,Fq", [(2, 37, 20), (3, 150, 20), (1, 70, 10)])
def test_attn2d_time_and_frequency_attention(dtype, B, T, Fq):
    """both attentions of conv_attention_2d.py:96-120 (no mask, no dropout), forward and backward, against torch"""
    H, M = 4, T * B * Fq
    qkv = torch.zeros(M, 16); qkv[:, :12] = torch.relu(rnd(M, 12, seed=1, scale=0.8))
    qkv = qkv.to(dtype)
    dcat = rnd(M, 8, dtype=dtype, seed=2)
    x = qkv.float().clone().requires_grad_(True)
    o_t, o_f = [], []
    for h in range(H):
        q, k, v = _planes(x, B, T, Fq, 16, h), _planes(x, B, T, Fq, 16, 4 + h), _planes(x, B, T, Fq, 16, 8 + h)
        o_t.append(torch.softmax(q @ k.transpose(1, 2), -1) @ v)
        o_f.append((torch.softmax(q.transpose(1, 2) @ k, -1) @ v.transpose(1, 2)).transpose(1, 2))
    ref = torch.stack(o_t + o_f, -1).permute(1, 0, 2, 3).reshape(M, 8)          # [B,T,F,8] -> rows (t,b,f)
    ref.backward(dcat.float())
    qd, dd = qkv.to(DEV), dcat.to(DEV)
    cat = torch.empty(M, 8, dtype=dtype, device=DEV)
    lse = K.a2d_time_fwd(qd, cat, B, T, Fq)
    A = K.a2d_freq_fwd(qd, cat, B, T, Fq)
    assert rel_err(cat, ref.detach()) < tol(dtype)
    assert abs(float(A.sum(-1).mean()) - 1.0) < 1e-5
    dq = torch.zeros_like(qd)
    K.a2d_time_bwd(qd, cat, dd, lse, dq, B, T, Fq)
    K.a2d_freq_bwd(qd, dd, A, dq, B, T, Fq)
    assert rel_err(dq[:, :12], x.grad[:, :12]) < 3 * tol(dtype)
    assert float(dq[:, 12:].abs().max()) == 0.0


def test_attn2d_dropout_is_an_unbiased_mask_and_backward_is_its_adjoint():
    B, T, Fq, M = 2, 64, 20, 2 * 64 * 20
    qkv = torch.zeros(M, 16); qkv[:, :12] = torch.relu(rnd(M, 12, seed=3, scale=0.5))
    qd = qkv.to(DEV)
    cat0 = torch.empty(M, 8, device=DEV); cat1 = torch.empty(M, 8, device=DEV); cat2 = torch.empty(M, 8, device=DEV)
    K.a2d_time_fwd(qd, cat0, B, T, Fq); K.a2d_freq_fwd(qd, cat0, B, T, Fq)
    lse = K.a2d_time_fwd(qd, cat1, B, T, Fq, 0.25, 11); A = K.a2d_freq_fwd(qd, cat1, B, T, Fq, 0.25, 12)
    K.a2d_time_fwd(qd, cat2, B, T, Fq, 0.25, 11); K.a2d_freq_fwd(qd, cat2, B, T, Fq, 0.25, 12)
    assert torch.equal(cat1, cat2) and not torch.equal(cat0, cat1)
    assert abs(float(cat1[:, :4].mean() / cat0[:, :4].mean()) - 1.0) < 0.02        # E[mask/(1-p)] = 1
    # the outputs are linear in v for a fixed mask: <dO, O(v)> = <dv, v>
    do = rnd(M, 8, seed=4).to(DEV)
    dq = torch.zeros_like(qd)
    K.a2d_time_bwd(qd, cat1, do, lse, dq, B, T, Fq, 0.25, 11)
    K.a2d_freq_bwd(qd, do, A, dq, B, T, Fq, 0.25, 12)
    lhs = float((do.double() * cat1.double()).sum()); rhs = float((dq[:, 8:12].double() * qd[:, 8:12].double()).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attn2d_batchnorm_relu_pieces(dtype):
    """grouped statistics -> s2t_bn_finalize -> BN + ReLU (+ residual) and its backward against torch batch_norm (training mode)"""
    M, C, Cg = 1500, 12, 4
    z = torch.zeros(M, 16); z[:, :C] = rnd(M, C, seed=5, scale=1.5) + 0.3
    z = z.to(dtype)
    ps = torch.tensor([0.125] * 4 + [1.0] * 12)
    gam, bet = 1 + 0.1 * rnd(C, seed=6), 0.1 * rnd(C, seed=7)
    dy = torch.zeros(M, 16); dy[:, :C] = rnd(M, C, seed=8); dy = dy.to(dtype)
    zf = z.float().clone().requires_grad_(True); gf = gam.clone().requires_grad_(True); bf = bet.clone().requires_grad_(True)
    zz = (zf * ps)[:, :C]
    ref = torch.relu(F.batch_norm(zz.t().reshape(1, C, M), None, None, gf, bf, True, 0.1, 1e-5)).reshape(C, M).t()
    ref.backward(dy.float()[:, :C])
    zd, psd = z.to(DEV), ps.to(DEV)
    sums = K.a2d_chan_stats(zd, C, Cg, prescale=psd)
    parts = []
    for gi in range(3):
        rm, rv = torch.zeros(Cg, device=DEV), torch.ones(Cg, device=DEV)
        parts.append(K.bn_finalize(sums[8 * gi:8 * gi + 8], gam[4 * gi:4 * gi + 4].to(DEV), bet[4 * gi:4 * gi + 4].to(DEV), rm, rv,
                                   torch.zeros(1, dtype=torch.int64, device=DEV), M, True))
        zg = zz[:, 4 * gi:4 * gi + 4].detach()
        assert rel_err(rm, 0.1 * zg.mean(0)) < 1e-3 and rel_err(rv, 0.9 + 0.1 * zg.var(0, unbiased=True)) < 1e-3
    bn = tuple(torch.cat([pt[k] for pt in parts] + [torch.zeros(4, device=DEV)]) for k in range(4))
    y = K.a2d_bn_act(zd, C, bn[2], bn[3], prescale=psd)
    assert rel_err(y[:, :C], ref.detach()) < tol(dtype) and float(y[:, C:].abs().max()) == 0.0
    s1 = K.a2d_chan_stats(zd, C, Cg, prescale=psd, dy=dy.to(DEV), bn=bn)
    dz = K.a2d_bn_bwd(dy.to(DEV), zd, C, Cg, bn, s1, M, True, prescale=psd)
    assert rel_err(dz[:, :C], zf.grad[:, :C]) < 3 * tol(dtype)
    dg, db = torch.zeros(Cg, device=DEV), torch.zeros(Cg, device=DEV)
    K.a2d_param_grads(s1[8:16], dg, db)
    assert rel_err(dg, gf.grad[4:8]) < (1e-3 if dtype == torch.float32 else 3e-2) and rel_err(db, bf.grad[4:8]) < (1e-3 if dtype == torch.float32 else 3e-2)


def test_attn2d_convolutions_as_gathered_gemms():
    """3x3 / pad 1 convolution forward, data gradient and weight gradient through the row maps + packed weights, against torch"""
    from fbk_fairseq_st_amd.engine import attn2d_maps
    B, T, Fq, Ci, Co = 2, 9, 20, 8, 64
    M = T * B * Fq
    x = rnd(M, Ci, seed=1); w = rnd(Co, Ci, 3, 3, seed=2, scale=0.3); bias = 0.1 * rnd(Co, seed=3); dy = rnd(M, Co, seed=4)
    xi = x.view(T, B, Fq, Ci).permute(1, 3, 0, 2).clone().requires_grad_(True)          # [B,Ci,T,F]
    wf = w.clone().requires_grad_(True)
    ref = F.conv2d(xi, wf, bias, padding=1)
    ref.backward(dy.view(T, B, Fq, Co).permute(1, 3, 0, 2))
    ref_rows = ref.detach().permute(2, 0, 3, 1).reshape(M, Co)
    mp = attn2d_maps(B, T, Fq, DEV)
    xd, dyd = x.to(DEV), dy.to(DEV)
    w0 = K.a2d_pack_w(w.to(DEV), Co, Ci, torch.float32, 0)
    y = K.gemm(xd, w0, M=M, K=9 * Ci, map_a=mp, period_a=Ci, bias=bias.to(DEV))
    assert rel_err(y, ref_rows) < 1e-4
    w1 = K.a2d_pack_w(w.to(DEV), Ci, Co, torch.float32, 1)
    dx = K.gemm(dyd, w1, M=M, K=9 * Co, map_a=mp, period_a=Co)
    assert rel_err(dx, xi.grad.permute(2, 0, 3, 1).reshape(M, Ci)) < 1e-4
    gp = torch.zeros(Co, 9 * Ci, device=DEV)
    for tap in range(9):
        K.gemm(dyd, xd, trans_a=True, trans_b=True, K=M, out=gp[:, tap * Ci:(tap + 1) * Ci], accumulate=True, splitk=2, map_b=mp[tap])
    gw = torch.zeros(Co, Ci, 3, 3, device=DEV)
    K.a2d_unpack_wgrad(gp, gw, Ci)
    assert rel_err(gw, wf.grad) < 1e-4
    gw1 = torch.ones(Co, Ci, 3, 3, device=DEV)                  # the one-pass kernel adds to the master layout
    assert K.a2d_conv_wgrad(dyd, xd, gw1, B, T, Fq)
    assert rel_err(gw1 - 1, wf.grad) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T", [(2, 9), (3, 40), (1, 6)])
def test_attn2d_in_projection_weight_gradient_one_pass(dtype, B, T):
    """12 real output channels stored with row stride 16, 64 input channels (the in_proj convolution)"""
    Fq, Ci, Co = 20, 64, 12
    M = T * B * Fq
    x = rnd(M, Ci, dtype=dtype, seed=1); dy16 = torch.zeros(M, 16); dy16[:, :Co] = rnd(M, Co, seed=2); dy16 = dy16.to(dtype)
    xi = x.float().view(T, B, Fq, Ci).permute(1, 3, 0, 2)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(xi, w, None, padding=1).backward(dy16.float()[:, :Co].reshape(T, B, Fq, Co).permute(1, 3, 0, 2))
    gw = torch.zeros(Co, Ci, 3, 3, device=DEV)
    assert K.a2d_conv_wgrad(dy16.to(DEV), x.to(DEV), gw, B, T, Fq)
    assert rel_err(gw, w.grad) < (1e-4 if dtype == torch.float32 else 1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_fused_into_bn_apply_and_scattered_gemm_epilogue(dtype):
    """the fused forms produce the bits of the separate s2t_dropout pass over the same tensor"""
    C, P = 64, 4096
    y = rnd(P, C, dtype=dtype, seed=1).to(DEV)
    sc, sh = (1 + 0.1 * rnd(C, seed=2)).to(DEV), (0.1 * rnd(C, seed=3)).to(DEV)
    assert torch.equal(K.bn_apply(y, sc, sh, 0.2, 99), K.dropout(K.bn_apply(y, sc, sh), 0.2, 99))
    # scattered output rows (map_c): the mask index is the element index of the destination tensor
    M, Kd, rows_out = 300, 128, 700
    a = rnd(M, Kd, dtype=dtype, seed=4).to(DEV); w = rnd(C, Kd, dtype=dtype, seed=5, scale=0.2).to(DEV)
    perm = torch.randperm(rows_out, generator=torch.Generator().manual_seed(6))[:M].to(torch.int32).to(DEV)
    o1 = torch.zeros(rows_out, C, dtype=dtype, device=DEV); o2 = torch.zeros(rows_out, C, dtype=dtype, device=DEV)
    K.gemm(a, w, map_c=perm, out=o1, p_drop=0.3, seed=7)
    K.gemm(a, w, map_c=perm, out=o2)
    ref = K.dropout(o2, 0.3, 7)
    assert torch.equal(o1 == 0, ref == 0)                       # the same mask ...
    if dtype == torch.float32:
        assert torch.equal(o1, ref)
    else:                                                       # ... the epilogue scales the f32 accumulator (one rounding instead of two)
        assert rel_err(o1, ref) < 1e-2

def test_attn_bwd_fused_matches_two_kernel_path():
    """s2t_set_option "attn_bwd_fused" (default off: measured no faster, profiles/r06_attn_bwd_fused.txt): the one-kernel attention backward
    for Tk <= 384 against the two-kernel path on the encoder's shape with dropout and ragged key lengths -- dV identical (the same
    arithmetic in the same order), dK / dQ within 4e-3 of the largest element (bf16 operands, another order over the keys)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    B, H, T = 3, 8, 375
    D = 64 * H
    qkv = (torch.randn(T, B, 3 * D, device=DEV, generator=g) * 0.7).to(torch.bfloat16)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    do = torch.randn(T, B, D, device=DEV, generator=g).to(torch.bfloat16)
    kl = torch.tensor([375, 250, 131], dtype=torch.int32, device=DEV)
    o, lse = K.attn_fwd(q, k, v, H, klen=kl, p_drop=0.1, seed=9)
    res = []
    try:
        for fused in (0, 1):
            K.set_option("attn_bwd_fused", fused)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            K.attn_bwd(q, k, v, o, do, lse, H, dq, dk, dv, klen=kl, p_drop=0.1, seed=9)
            torch.cuda.synchronize()
            res.append((dq.float(), dk.float(), dv.float()))
    finally:
        K.set_option("attn_bwd_fused", 0)
    (q0, k0, v0), (q1, k1, v1) = res
    assert torch.equal(v0, v1)
    for a, b_ in ((q0, q1), (k0, k1)):
        assert torch.isfinite(b_).all()
        assert float((a - b_).abs().max()) <= 4e-3 * float(a.abs().max())
