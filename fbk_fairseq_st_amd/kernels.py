"""Thin tensor-level wrappers over the C ABI (one function per entry point of include/s2t_hip.h).

No arithmetic happens here: each wrapper validates shapes, allocates outputs with torch (device
memory is torch's job) and passes raw pointers to libs2t_hip.so on torch's current stream.
"""
import ctypes

import torch

from . import lib as L
from .lib import ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_BWD, ACT_GELU_BWD, ACT_RELU_MASK, ACT_RELU_BWD_MASK  # noqa: F401


def _lib():
    return L.load()


_GEMM = None
_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}
_RAW_STREAM = torch._C._cuda_getCurrentRawStream


def gemm(a, b, trans_a=False, trans_b=False, bias=None, residual=None, act=ACT_NONE, aux=None, aux_out=None,
         out=None, out_dtype=None, accumulate=False, splitk=1, alpha=1.0, M=None, N=None, K=None,
         map_a=None, period_a=0, map_b=None, map_c=None, out_rows=None, p_drop=0.0, seed=0):
    """C = epi(op(a) @ op(b)); a is [M,K] (or [K,M] if trans_a), b is [N,K] (or [K,N] if trans_b).

    The most frequent call of an update (~200 per step), written for host time: the bound C function and the dtype codes are
    looked up once, pointers are taken inline (a host tensor would be a GPU fault, so the operands are still checked)."""
    global _GEMM
    if _GEMM is None:
        _GEMM = _lib().s2t_gemm_gather
    if not (a.is_cuda and b.is_cuda):
        raise L.S2THipError("S2T kernels need device tensors: the hot path has no CPU fallback")
    sa0, sb0 = a.stride(0), b.stride(0)
    assert a.stride(1) == 1 and b.stride(1) == 1
    if M is None:
        M = a.shape[1] if trans_a else a.shape[0]
    if K is None:
        K = a.shape[0] if trans_a else a.shape[1]
    if N is None:
        N = b.shape[1] if trans_b else b.shape[0]
    if out is None:
        odt = out_dtype or a.dtype
        rows = out_rows if out_rows is not None else M
        out = (torch.zeros if (accumulate or splitk > 1 or map_c is not None) else torch.empty)(
            (rows, N), dtype=odt, device=a.device)
    ldaux = 0
    if act >= ACT_RELU_MASK:                 # aux / aux_out is the 1-bit record (relu_mask_bytes), not a tensor of the output's shape
        t = aux_out if act == ACT_RELU_MASK else aux
        assert t is not None and t.is_cuda and t.dtype == torch.uint8 and t.numel() == relu_mask_bytes(M, N, K) > 0
    elif residual is not None or aux is not None or aux_out is not None:
        for t in (residual, aux, aux_out):
            assert t is None or (t.dtype == out.dtype and t.stride(-1) == 1 and t.is_cuda)
        ldaux = aux.stride(0) if aux is not None else (aux_out.stride(0) if aux_out is not None else 0)
    rc = _GEMM(
        _DT[a.dtype], _DT[out.dtype], int(trans_a), int(trans_b), M, N, K, a.data_ptr(), sa0, b.data_ptr(), sb0,
        out.data_ptr(), out.stride(0), bias.data_ptr() if bias is not None else 0,
        residual.data_ptr() if residual is not None else 0, residual.stride(0) if residual is not None else 0,
        aux.data_ptr() if aux is not None else 0, aux_out.data_ptr() if aux_out is not None else 0, ldaux, act,
        int(accumulate), splitk, alpha,
        L.ptr(map_a), period_a, L.ptr(map_b), L.ptr(map_c), p_drop, seed, _RAW_STREAM(L.device_index()))
    if rc:
        L.check(rc, "s2t_gemm")
    return out


_MASK_BYTES = {}


def relu_mask_bytes(M, N, K):
    """Bytes of the 1-bit ReLU record of an [M, N] bf16 product over K (ACT_RELU_MASK / ACT_RELU_BWD_MASK); 0 = products of this
    shape keep the activations as the backward operand (ACT_RELU / ACT_RELU_BWD)."""
    key = (M, N, K, OPTION_EPOCH)           # the tile height, hence the record's size, follows the route options (reserve_cus, gemm256*)
    n = _MASK_BYTES.get(key)
    if n is None:
        if len(_MASK_BYTES) > 4096:
            _MASK_BYTES.clear()
        n = _MASK_BYTES[key] = int(_lib().s2t_gemm_relu_mask_bytes(M, N, K))
    return n


def linear_wgrad(dy, x, dw, db=None, splitk=1):
    """dw[n_out,n_in] += dy[tokens,n_out]^T x[tokens,n_in]; db[n_out] += column sums of dy (one pass over dy)."""
    L.require_cuda(dy, x)
    assert dy.dim() == 2 and x.dim() == 2 and dy.stride(1) == 1 and x.stride(1) == 1 and dy.shape[0] == x.shape[0]
    assert dw.dtype == torch.float32 and dw.stride(1) == 1 and (db is None or db.dtype == torch.float32)
    L.check(_lib().s2t_linear_wgrad(L.dt(dy), dy.shape[1], x.shape[1], dy.shape[0], L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0),
                                    L.ptr(dw), dw.stride(0), L.ptr(db), int(splitk), L.stream()), "s2t_linear_wgrad")
    return dw


WGRAD_GROUP_MAX = 4096


def wgrad_group(items):
    """items: list of (dy [tokens, n_out], x [tokens, n_in], dw [n_out, n_in] f32, db [n_out] f32 or None), bf16 operands:
    dw += dy^T x and db += column sums of dy for all of them in one launch (s2t_wgrad_group).  The operands were checked by
    wgrad_group_ok when they were queued; here only what the C side cannot see is re-checked (host time: this runs at the end of
    the decoder's backward, where the launch stream is what the GPU waits for)."""
    if not items:
        return
    fn = _lib().s2t_wgrad_group if items[0][0].dtype == torch.bfloat16 else _lib().s2t_wgrad_group_f32
    for i in range(0, len(items), WGRAD_GROUP_MAX):
        chunk = items[i:i + WGRAD_GROUP_MAX]
        arr = (L.WgradProblem * len(chunk))()
        for k, (dy, x, dw, db) in enumerate(chunk):
            if not (dy.is_cuda and x.is_cuda and dw.is_cuda and dw.dtype == torch.float32 and dy.shape[0] == x.shape[0]
                    and dw.shape[0] == dy.shape[1] and dw.shape[1] == x.shape[1] and dy.dtype == x.dtype == items[0][0].dtype):
                raise L.S2THipError("wgrad_group: item %d is not a (dy [tokens, n_out], x [tokens, n_in], f32 dw [n_out, n_in]) device triple" % k)
            p = arr[k]
            p.dY = dy.data_ptr(); p.X = x.data_ptr(); p.dW = dw.data_ptr(); p.db = db.data_ptr() if db is not None else None
            p.n_out = dy.shape[1]; p.n_in = x.shape[1]; p.tokens = dy.shape[0]
            p.ldy = dy.stride(0); p.ldx = x.stride(0); p.ldw = dw.stride(0)
        L.check(fn(len(chunk), ctypes.addressof(arr), L.stream()), "s2t_wgrad_group")


def wgrad_group_raw(n, items_addr):
    """launch n S2TWgradProblem entries of a host array the caller filled (engine: the products its layer calls appended)"""
    if n:
        L.check(_lib().s2t_wgrad_group(int(n), int(items_addr), L.stream()), "s2t_wgrad_group")


def layer_ws_bytes(desc, training):
    return int(_lib().s2t_layer_ws_bytes(ctypes.addressof(desc), int(training)))


def layer_tmp_bytes(desc):
    return int(_lib().s2t_layer_bwd_tmp_bytes(ctypes.addressof(desc)))


def layer_fwd(desc_addr, call_addr):
    rc = _lib().s2t_layer_fwd(desc_addr, call_addr, L.stream())
    if rc:
        L.check(rc, "s2t_layer_fwd")


def layer_bwd(desc_addr, call_addr):
    rc = _lib().s2t_layer_bwd(desc_addr, call_addr, L.stream())
    if rc:
        L.check(rc, "s2t_layer_bwd")


def wgrad_group_ok(dy, x):
    """shapes the grouped kernels take (otherwise: linear_wgrad): bf16 -> s2t_wgrad_group, f32 -> s2t_wgrad_group_f32"""
    if dy.dtype == torch.float32 and x.dtype == torch.float32:
        return (dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0
                and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
                and (dy.shape[1] + 3) // 4 * 4 <= dy.stride(0) and (x.shape[1] + 3) // 4 * 4 <= x.stride(0))
    return (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.stride(1) == 1 and x.stride(1) == 1
            and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
            and dy.shape[1] >= 8 and x.shape[1] >= 8
            and (dy.shape[1] + 7) // 8 * 8 <= dy.stride(0) and (x.shape[1] + 7) // 8 * 8 <= x.stride(0))


def colsum(x, out):
    """out[n] += sum_m x[m, n] (f32)."""
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32
    L.check(_lib().s2t_colsum(L.dt(x), L.ptr(x), x.stride(0), x.shape[0], x.shape[1], L.ptr(out), L.stream()), "s2t_colsum")
    return out


def _tb(x):
    """(time stride, batch stride) in elements of a [T,B,...] view whose last dim is contiguous."""
    return x.stride(0), x.stride(1)


def attn_fwd(q, k, v, heads, klen=None, causal=False, scale=None, p_drop=0.0, seed=0, out=None, dist_penalty=False):
    """q [Tq,B,D'], k/v [Tk,B,D'] views (last dim contiguous, may be slices of a fused QKV buffer)."""
    Tq, B, D = q.shape
    Tk = k.shape[0]
    d = D // heads
    if scale is None:
        scale = d ** -0.5
    if out is None:
        out = torch.empty((Tq, B, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, heads, Tq), dtype=torch.float32, device=q.device)
    rc = _lib().s2t_attn_fwd(L.dt(q), d, B, heads, Tq, Tk, L.ptr(q), *_tb(q), L.ptr(k), *_tb(k), L.ptr(v), *_tb(v),
                             L.ptr(out), *_tb(out), L.ptr(lse), L.ptr(klen), int(causal), int(bool(dist_penalty)), float(scale), float(p_drop),
                             int(seed), L.stream())
    L.check(rc, "s2t_attn_fwd")
    return out, lse


def attn_probs_avg(q, k, heads, klen=None, scale=None, heads_used=None):
    """head-averaged attention probabilities of one layer, recomputed from q [Tq,B,D'] and k [Tk,B,D'] -> f32 [B, Tq, Tk]
    (transformer.py:756-782: what the decoder returns as `attn` for its alignment layer)"""
    Tq, B, D = q.shape
    Tk = k.shape[0]
    d = D // heads
    out = torch.empty((B, Tq, Tk), dtype=torch.float32, device=q.device)
    L.check(_lib().s2t_attn_probs_avg(L.dt(q), d, B, heads, Tq, Tk, L.ptr(q), *_tb(q), L.ptr(k), *_tb(k), L.ptr(klen),
                                      int(heads_used or heads), float(d ** -0.5 if scale is None else scale), L.ptr(out), L.stream()),
            "s2t_attn_probs_avg")
    return out


def attn_bwd(q, k, v, o, do, lse, heads, dq, dk, dv, klen=None, causal=False, scale=None, p_drop=0.0, seed=0, dist_penalty=False):
    Tq, B, D = q.shape
    Tk = k.shape[0]
    d = D // heads
    if scale is None:
        scale = d ** -0.5
    delta = torch.empty((B, heads, Tq), dtype=torch.float32, device=q.device)
    rc = _lib().s2t_attn_bwd(L.dt(q), d, B, heads, Tq, Tk, L.ptr(q), *_tb(q), L.ptr(k), *_tb(k), L.ptr(v), *_tb(v),
                             L.ptr(o), *_tb(o), L.ptr(do), *_tb(do), L.ptr(lse), L.ptr(delta),
                             L.ptr(dq), *_tb(dq), L.ptr(dk), *_tb(dk), L.ptr(dv), *_tb(dv),
                             L.ptr(klen), int(causal), int(bool(dist_penalty)), float(scale), float(p_drop), int(seed), L.stream())
    L.check(rc, "s2t_attn_bwd")
    return dq, dk, dv


_LN_FWD = _LN_BWD = None


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    """-> (y, mean, rstd).  88 calls per update, written for host time like gemm(): one allocation for both statistics, the bound C
    function looked up once, pointers taken inline (the operands are still checked: a host pointer would be a GPU fault)."""
    global _LN_FWD
    if _LN_FWD is None:
        _LN_FWD = _lib().s2t_layernorm_fwd
    if not (x.is_cuda and gamma.is_cuda and beta.is_cuda):
        raise L.S2THipError("S2T kernels need device tensors: the hot path has no CPU fallback")
    D = x.shape[-1]
    M = x.numel() // D
    assert x.is_contiguous()
    y = torch.empty_like(x)
    st = torch.empty((2, M), dtype=torch.float32, device=x.device)
    mean, rstd = st[0], st[1]
    rc = _LN_FWD(_DT[x.dtype], x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                 M, D, float(eps), _RAW_STREAM(L.device_index()))
    if rc:
        L.check(rc, "s2t_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dres=None, drop=None):
    """drop = (p, seed) of the dropout that consumes dx next: returns (dx, dropout(dx, p, seed)) from one pass."""
    global _LN_BWD
    if _LN_BWD is None:
        _LN_BWD = _lib().s2t_layernorm_bwd
    if not (dy.is_cuda and x.is_cuda and mean.is_cuda and rstd.is_cuda and gamma.is_cuda and dgamma.is_cuda and dbeta.is_cuda
            and (dres is None or dres.is_cuda)):
        raise L.S2THipError("S2T kernels need device tensors: the hot path has no CPU fallback")
    D = x.shape[-1]
    M = x.numel() // D
    assert dy.is_contiguous() and x.is_contiguous() and (dres is None or dres.is_contiguous())
    dx = torch.empty_like(x)
    dxd = torch.empty_like(x) if drop is not None else None
    p, seed = drop if drop is not None else (0.0, 0)
    rc = _LN_BWD(_DT[x.dtype], dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                 dres.data_ptr() if dres is not None else 0, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), M, D,
                 dxd.data_ptr() if dxd is not None else 0, float(p), int(seed), _RAW_STREAM(L.device_index()))
    if rc:
        L.check(rc, "s2t_layernorm_bwd")
    return dx if drop is None else (dx, dxd)


def conv1_fwd(x, w, bias, C, dtype, act=ACT_RELU):
    """-> (y, sums, pre): pre = the pre-activation for GELU (None for ReLU, whose backward reads the mask off y)"""
    B, T, F = x.shape
    T2, F2 = (T + 1) // 2, (F + 1) // 2
    y = torch.empty((B, T2, F2, C), dtype=dtype, device=x.device)
    pre = torch.empty_like(y) if act == ACT_GELU else None
    sums = torch.zeros((2 * C,), dtype=torch.float64, device=x.device)
    L.check(_lib().s2t_conv1_fwd(L.dt(y), L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), L.ptr(pre), L.ptr(sums), B, T, F, C, act,
                                 L.stream()), "s2t_conv1_fwd")
    return y, sums, pre


def conv1_bwd(x, dpre, dw, db):
    B, T, F = x.shape
    C = dpre.shape[-1]
    L.check(_lib().s2t_conv1_bwd(L.dt(dpre), L.ptr(x), L.ptr(dpre), L.ptr(dw), L.ptr(db), B, T, F, C, L.stream()),
            "s2t_conv1_bwd")


def chan_sums(y, C, dyn=None, mean=None, rstd=None):
    P = y.numel() // C
    sums = torch.zeros((2 * C,), dtype=torch.float64, device=y.device)
    L.check(_lib().s2t_chan_sums(L.dt(y), L.ptr(y), L.ptr(dyn), L.ptr(mean), L.ptr(rstd), L.ptr(sums), P, C,
                                 0 if dyn is None else 1, L.stream()), "s2t_chan_sums")
    return sums


def conv1_bwd_bn(x, dyn, y, mean, rstd, gamma, sums, dw, db, dgamma, dbeta, count, training=True, pre=None):
    """bn_bwd_apply + conv1_bwd in one pass: the gradient w.r.t. conv1's output stays in registers"""
    B, T, F = x.shape
    C = y.shape[-1]
    L.check(_lib().s2t_conv1_bwd_bn(L.dt(y), L.ptr(x), L.ptr(dyn), L.ptr(y), L.ptr(pre), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(sums),
                                    L.ptr(dw), L.ptr(db), L.ptr(dgamma), L.ptr(dbeta), B, T, F, C, float(count), int(training), L.stream()),
            "s2t_conv1_bwd_bn")


def bn_finalize(sums, gamma, beta, run_mean, run_var, num_batches, count, training, momentum=0.1, eps=1e-5):
    C = gamma.numel()
    o = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    L.check(_lib().s2t_bn_finalize(L.ptr(sums), L.ptr(gamma), L.ptr(beta), L.ptr(run_mean), L.ptr(run_var),
                                   L.ptr(num_batches), L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.ptr(o[3]),
                                   float(count), C, int(training), float(momentum), float(eps), L.stream()), "s2t_bn_finalize")
    return o[0], o[1], o[2], o[3]            # mean, rstd, scale, shift


def bn_apply(y, scale, shift, p_drop=0.0, seed=0):
    """yn = dropout(y*scale + shift, p_drop, seed) in one pass (same bits as dropout(bn_apply(y)))"""
    yn = torch.empty_like(y)
    L.check(_lib().s2t_bn_apply(L.dt(y), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(yn), y.numel(), scale.numel(), float(p_drop),
                                int(seed), L.stream()), "s2t_bn_apply")
    return yn


def bn_bwd_apply(dyn, y, mean, rstd, gamma, sums, dgamma, dbeta, count, training=True, pre=None):
    dpre = torch.empty_like(y)
    L.check(_lib().s2t_bn_bwd_apply(L.dt(y), L.ptr(dyn), L.ptr(y), L.ptr(pre), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(sums),
                                    L.ptr(dpre), L.ptr(dgamma), L.ptr(dbeta), y.numel(), gamma.numel(), float(count),
                                    int(training), L.stream()), "s2t_bn_bwd_apply")
    return dpre


def permute_cf(src, dst, N, C, F, mode):
    L.check(_lib().s2t_permute_cf(L.dt(dst), L.ptr(src), L.ptr(dst), N, C, F, mode, L.stream()), "s2t_permute_cf")
    return dst


def permute_conv_w(src, dst, Co, Ci, mode):
    L.check(_lib().s2t_permute_conv_w(L.dt(dst), L.ptr(src), L.ptr(dst), Co, Ci, mode, L.stream()), "s2t_permute_conv_w")
    return dst


def add_pos(x, table, len32, out=None, p_drop=0.0, seed=0):
    """out = dropout(x + positions) in one pass (out = None: in place)"""
    T, B, D = x.shape
    out = x if out is None else out
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape and out.dtype == x.dtype
    assert table.shape[0] >= T + 1 and table.shape[1] == D
    L.check(_lib().s2t_add_pos(L.dt(x), L.ptr(x), L.ptr(out), L.ptr(table), L.ptr(len32), T, B, D, float(p_drop), int(seed), L.stream()),
            "s2t_add_pos")
    return out


def padded_cols(V, dtype):
    """row stride (elements) that keeps every row 16-byte aligned"""
    e = 8 if dtype == torch.bfloat16 else 4
    return (V + e - 1) // e * e


def alloc_rows(rows_shape, V, dtype, device, zero=False):
    """[..., V] view of a buffer whose row stride is padded (see padded_cols)."""
    ld = padded_cols(V, dtype)
    buf = (torch.zeros if zero else torch.empty)(tuple(rows_shape) + (ld,), dtype=dtype, device=device)
    return buf[..., :V]


def _row_ld(x):
    """row stride of a [T,B,V] / [rows,V] view with dense leading dims"""
    assert x.stride(-1) == 1
    ld = x.stride(-2)
    if x.dim() == 3:
        if x.shape[1] == 1:                    # batch of one: the stride torch reports for a size-1 dimension is arbitrary
            return x.stride(0) if x.shape[0] > 1 else max(ld, x.shape[2])
        assert x.shape[0] == 1 or x.stride(0) == x.shape[1] * ld
    return ld


def ctc_argmax(logits, want_lse=False):
    """(pred, pmax) [B, T]; want_lse: also the rows' log-sum-exps [T*B] (same pass), which ctc_loss(lse=...) reuses"""
    T, B, V = logits.shape
    pred = torch.empty((B, T), dtype=torch.int32, device=logits.device)
    pmax = torch.empty((B, T), dtype=torch.float32, device=logits.device)
    lse = torch.empty((T * B,), dtype=torch.float32, device=logits.device) if want_lse else None
    L.check(_lib().s2t_ctc_argmax(L.dt(logits), L.ptr(logits), L.ptr(pred), L.ptr(pmax), L.ptr(lse), T, B, V, _row_ld(logits), L.stream()),
            "s2t_ctc_argmax")
    return (pred, pmax, lse) if want_lse else (pred, pmax)


def ctc_rle(pred, pmax, len64, strategy=0):
    B, T = pred.shape
    dev = pred.device
    seg = torch.empty((B, T), dtype=torch.int32, device=dev)
    rs = torch.empty((B, T), dtype=torch.int32, device=dev)
    rl = torch.empty((B, T), dtype=torch.int32, device=dev)
    new_len = torch.empty((B,), dtype=torch.int64, device=dev)
    w = torch.empty((B, T), dtype=torch.float32, device=dev)
    L.check(_lib().s2t_ctc_rle(L.ptr(pred), L.ptr(pmax), L.ptr(len64), L.ptr(seg), L.ptr(rs), L.ptr(rl), L.ptr(new_len),
                               L.ptr(w), T, B, strategy, L.stream()), "s2t_ctc_rle")
    return seg, rs, rl, new_len, w


def ctc_compress_fwd(x, w, rs, rl, new_len, Tout):
    T, B, D = x.shape
    out = torch.empty((Tout, B, D), dtype=x.dtype, device=x.device)
    L.check(_lib().s2t_ctc_compress_fwd(L.dt(x), L.ptr(x), L.ptr(w), L.ptr(rs), L.ptr(rl), L.ptr(new_len), L.ptr(out),
                                        T, B, D, Tout, L.stream()), "s2t_ctc_compress_fwd")
    return out


def ctc_compress_bwd(dout, w, seg, dx, accumulate=False):
    T, B, D = dx.shape
    L.check(_lib().s2t_ctc_compress_bwd(L.dt(dx), L.ptr(dout), L.ptr(w), L.ptr(seg), L.ptr(dx), T, B, D, int(accumulate),
                                        L.stream()), "s2t_ctc_compress_bwd")
    return dx


CTC_MAX_TARGET, CTC_MAX_VOCAB = 511, 40704        # S2T_CTC_MAX_TARGET / S2T_CTC_MAX_VOCAB of include/s2t_hip.h


def ctc_loss(logits, targets, tgt_len, in_len32, blank, grad_scale=1.0, defer_grad=False, lse=None):
    """Returns (loss_sum f32[1], grad like logits, nll).  defer_grad: the forward pass only; the second result is then the workspace
    tuple for ctc_loss_grad (called from backward with the upstream gradient as a device scalar: no separate scaling pass)."""
    T, B, V = logits.shape
    Lmax = targets.shape[1]
    if Lmax > CTC_MAX_TARGET or V > CTC_MAX_VOCAB:
        raise L.S2THipError("CTC loss kernels take transcripts of at most %d units and vocabularies of at most %d entries "
                            "(S2T_CTC_MAX_TARGET / S2T_CTC_MAX_VOCAB, include/s2t_hip.h); this batch has %d / %d: filter the data "
                            "with --max-target-positions or shorten the transcripts" % (CTC_MAX_TARGET, CTC_MAX_VOCAB, Lmax, V))
    dev = logits.device
    S = next(r for r in (64, 128, 256, 512, 1024) if 2 * Lmax + 1 <= r)      # S2T_CTC_ROW(Lmax), include/s2t_hip.h
    lse_given = lse is not None                     # row log-sum-exps of THESE logits from ctc_argmax(want_lse=True)
    if lse is None:
        lse = torch.empty((T * B,), dtype=torch.float32, device=dev)
    assert lse.numel() == T * B and lse.dtype == torch.float32
    la = torch.empty((B * T * S,), dtype=torch.float32, device=dev)
    lb = torch.empty((B * T * S,), dtype=torch.float32, device=dev)
    nll = torch.empty((B,), dtype=torch.float32, device=dev)
    ld = _row_ld(logits)
    grad = None if defer_grad else torch.empty((T, B, ld), dtype=logits.dtype, device=dev)[..., :V]
    loss = torch.zeros((1,), dtype=torch.float32, device=dev)
    L.check(_lib().s2t_ctc_loss(L.dt(logits), L.ptr(logits), L.ptr(targets), L.ptr(tgt_len), L.ptr(in_len32), L.ptr(lse),
                                L.ptr(la), L.ptr(lb), L.ptr(nll), L.ptr(grad), L.ptr(loss), T, B, V, ld, Lmax, blank,
                                float(grad_scale), (1 if defer_grad else 0) | (4 if lse_given else 0), 0, L.stream()), "s2t_ctc_loss")
    if defer_grad:
        return loss, (logits, targets, tgt_len, in_len32, lse, la, lb, nll, blank, float(grad_scale)), nll
    return loss, grad, nll


def ctc_loss_grad(ws, upstream):
    """gradient w.r.t. the logits from the workspaces of ctc_loss(defer_grad=True), times the device scalar `upstream` (f32[1])"""
    logits, targets, tgt_len, in_len32, lse, la, lb, nll, blank, grad_scale = ws
    T, B, V = logits.shape
    ld = _row_ld(logits)
    grad = torch.empty((T, B, ld), dtype=logits.dtype, device=logits.device)[..., :V]
    assert upstream.dtype == torch.float32 and upstream.numel() == 1
    L.check(_lib().s2t_ctc_loss(L.dt(logits), L.ptr(logits), L.ptr(targets), L.ptr(tgt_len), L.ptr(in_len32), L.ptr(lse),
                                L.ptr(la), L.ptr(lb), L.ptr(nll), L.ptr(grad), 0, T, B, V, ld, targets.shape[1], blank,
                                grad_scale, 2, L.ptr(upstream), L.stream()), "s2t_ctc_loss")
    return grad


def lsce(logits, target, eps, pad, want_grad=True, grad_scale=1.0):
    """logits [rows,V]; returns (sums f32[2] = loss, nll ; dlogits or None)."""
    rows, V = logits.shape
    assert target.numel() == rows
    ld = _row_ld(logits)
    sums = torch.zeros((2,), dtype=torch.float32, device=logits.device)
    dl = torch.empty((rows, ld), dtype=logits.dtype, device=logits.device)[:, :V] if want_grad else None
    L.check(_lib().s2t_lsce(L.dt(logits), L.ptr(logits), L.ptr(target), L.ptr(dl), L.ptr(sums), rows, V, ld, float(eps), pad,
                            float(grad_scale), L.stream()), "s2t_lsce")
    return sums, dl


def kd_loss(logits, target, teacher_idx, teacher_logits, lam, tau, pad, want_grad=True, grad_scale=1.0):
    """logits [rows,V]; teacher_idx/logits [rows,Kt]. Returns (loss f32[1], dlogits or None)."""
    rows, V = logits.shape
    ld = _row_ld(logits)
    s = torch.zeros((1,), dtype=torch.float32, device=logits.device)
    dl = torch.empty((rows, ld), dtype=logits.dtype, device=logits.device)[:, :V] if want_grad else None
    Kt = teacher_idx.shape[-1] if teacher_idx is not None else 0
    L.check(_lib().s2t_kd_loss(L.dt(logits), L.ptr(logits), L.ptr(target), L.ptr(teacher_idx), L.ptr(teacher_logits), L.ptr(dl),
                               L.ptr(s), rows, V, ld, Kt, float(lam), float(tau), pad, float(grad_scale), L.stream()), "s2t_kd_loss")
    return s, dl


def embed_fwd(tokens, W, table, scale, pad, pos_offset=0):
    B, Ln = tokens.shape
    D = W.shape[1]
    out = torch.empty((Ln, B, D), dtype=W.dtype, device=W.device)
    L.check(_lib().s2t_embed_fwd(L.dt(W), L.ptr(tokens), L.ptr(W), L.ptr(table), L.ptr(out), B, Ln, D, float(scale), pad,
                                 int(pos_offset), L.stream()), "s2t_embed_fwd")
    return out


def conv2_wgrad(dpre, y1n, gw, B, T2, F2, C):
    """all-taps conv2 weight gradient; returns False when the shape / dtype is not covered (caller uses the gathered GEMMs)"""
    rc = _lib().s2t_conv2_wgrad(L.dt(dpre), L.ptr(dpre), L.ptr(y1n), L.ptr(gw), B, T2, F2, C, L.stream())
    if rc == -95:
        return False
    L.check(rc, "s2t_conv2_wgrad")
    return True


def conv2_fwd(y1n, w2p, bias, B, T2, F2, C, act=ACT_RELU):
    """direct stride-2 3x3 convolution of the subsampler on channels-last rows: -> (z2 [T4*B*F4, C], pre or None), or None when the
    shape / dtype is not covered (caller uses the gathered GEMM)"""
    T4, F4 = (T2 + 1) // 2, (F2 + 1) // 2
    z2 = torch.empty((T4 * B * F4, C), dtype=y1n.dtype, device=y1n.device)
    pre = torch.empty_like(z2) if act == ACT_GELU else None
    rc = _lib().s2t_conv2_fwd(L.dt(y1n), L.ptr(y1n), L.ptr(w2p), L.ptr(bias), L.ptr(z2), L.ptr(pre), B, T2, F2, C, act, L.stream())
    if rc == -95:
        return None
    L.check(rc, "s2t_conv2_fwd")
    return z2, pre


def conv2_dgrad(dpre, w2q, dy1n, B, T2, F2, C, p_drop=0.0, seed=0):
    """data gradient of the stride-2 3x3 convolution into dy1n [B*T2*F2, C] (every element written), dropout mask of y1n applied;
    False when the shape / dtype is not covered (caller uses the four gathered products)"""
    rc = _lib().s2t_conv2_dgrad(L.dt(dpre), L.ptr(dpre), L.ptr(w2q), L.ptr(dy1n), B, T2, F2, C, float(p_drop), int(seed), L.stream())
    if rc == -95:
        return False
    L.check(rc, "s2t_conv2_dgrad")
    return True


def topk(logits, k):
    """[rows,V] (row stride may be padded) -> (f32 [rows,k] values descending, int32 [rows,k] columns)"""
    rows, V = logits.shape
    vals = torch.empty((rows, k), dtype=torch.float32, device=logits.device)
    idx = torch.empty((rows, k), dtype=torch.int32, device=logits.device)
    L.check(_lib().s2t_topk(L.dt(logits), L.ptr(logits), L.ptr(vals), L.ptr(idx), rows, V, _row_ld(logits), int(k), L.stream()), "s2t_topk")
    return vals, idx


def log_softmax(logits, temperature=1.0):
    """[rows,V] (row stride may be padded) -> f32 [rows,V] log-probabilities"""
    rows, V = logits.shape
    out = torch.empty((rows, V), dtype=torch.float32, device=logits.device)
    L.check(_lib().s2t_log_softmax(L.dt(logits), L.ptr(logits), L.ptr(out), rows, V, _row_ld(logits), 1.0 / float(temperature),
                                   L.stream()), "s2t_log_softmax")
    return out


def softmax_probs(logits, temperature=1.0):
    """[rows,V] (row stride may be padded) -> f32 [rows,V] probabilities"""
    rows, V = logits.shape
    out = torch.empty((rows, V), dtype=torch.float32, device=logits.device)
    L.check(_lib().s2t_softmax_probs(L.dt(logits), L.ptr(logits), L.ptr(out), rows, V, _row_ld(logits), 1.0 / float(temperature),
                                     L.stream()), "s2t_softmax_probs")
    return out


def softmax_bwd(out, dout, dtype, log_probs, temperature=1.0):
    """gradient of log_softmax / softmax_probs w.r.t. the logits from the saved f32 output and its gradient -> [rows,V] of `dtype`"""
    rows, V = out.shape
    assert out.dtype == torch.float32 and dout.dtype == torch.float32 and out.is_contiguous() and dout.is_contiguous()
    dx = torch.empty((rows, V), dtype=dtype, device=out.device)
    L.check(_lib().s2t_softmax_bwd(L.dt(dx), L.ptr(out), L.ptr(dout), L.ptr(dx), rows, V, V, 1.0 / float(temperature), int(bool(log_probs)),
                                   L.stream()), "s2t_softmax_bwd")
    return dx


class NormalizedProbs(torch.autograd.Function):
    """log_softmax / softmax of logit rows with a gradient (get_normalized_probs for criteria that consume the (log-)probabilities:
    fairseq/models/fairseq_decoder.py:58-79); both directions are kernels of csrc/loss_embed.hip"""

    @staticmethod
    def forward(ctx, logits2d, log_probs):
        out = log_softmax(logits2d) if log_probs else softmax_probs(logits2d)
        ctx.save_for_backward(out)
        ctx.log_probs, ctx.dtype = bool(log_probs), logits2d.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        return softmax_bwd(out, dout.contiguous().float(), ctx.dtype, ctx.log_probs), None


def ensemble_lse(lprobs):
    """list of f32 [rows, V] log-probabilities (one per ensemble member) -> log of their mean probability, element-wise"""
    n = len(lprobs)
    assert 1 <= n <= 8 and all(t.dtype == torch.float32 and t.is_contiguous() and t.shape == lprobs[0].shape for t in lprobs)
    out = torch.empty_like(lprobs[0])
    arr = (ctypes.c_void_p * n)(*[L.ptr(t) for t in lprobs])
    L.check(_lib().s2t_ensemble_lse(n, ctypes.addressof(arr), L.ptr(out), out.numel(), L.stream()), "s2t_ensemble_lse")
    return out


def embed_bwd(tokens, dout, dW, scale, pad):
    B, Ln = tokens.shape
    D = dW.shape[1]
    L.check(_lib().s2t_embed_bwd(L.dt(dout), L.ptr(tokens), L.ptr(dout), L.ptr(dW), B, Ln, D, float(scale), pad, L.stream()),
            "s2t_embed_bwd")


def act_bwd(dy, y, act, p_drop=0.0, seed=0):
    """dropout(dy) * act'(y): the backward of act followed by a dropout, one pass"""
    assert dy.is_contiguous() and y.is_contiguous() and dy.dtype == y.dtype and dy.numel() == y.numel()
    out = torch.empty_like(dy)
    L.check(_lib().s2t_act_bwd(L.dt(dy), L.ptr(dy), L.ptr(y), L.ptr(out), dy.numel(), act, float(p_drop), int(seed), L.stream()), "s2t_act_bwd")
    return out


def add_inplace(x, y):
    """y += x"""
    assert x.numel() == y.numel() and x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous()
    L.check(_lib().s2t_add_inplace(L.dt(y), L.ptr(x), L.ptr(y), y.numel(), L.stream()), "s2t_add_inplace")
    return y


def dropout(x, p, seed, out=None):
    if out is None:
        out = torch.empty_like(x)
    L.check(_lib().s2t_dropout(L.dt(x), L.ptr(x), L.ptr(out), x.numel(), float(p), int(seed), L.stream()), "s2t_dropout")
    return out


def grad_norm_clip(g, scale, max_norm, ws, out2, divisor=None):
    """divisor: f64 device scalar the gradients are also divided by (max(divisor, 1)): the all-reduced sample size, never read by the host"""
    if divisor is not None:
        assert divisor.dtype == torch.float64 and divisor.numel() == 1
        L.check(_lib().s2t_grad_norm_clip_div(L.ptr(g), g.numel(), L.ptr(ws), float(scale), L.ptr(divisor), float(max_norm), L.ptr(out2),
                                              L.stream()), "s2t_grad_norm_clip_div")
        return out2
    L.check(_lib().s2t_grad_norm_clip(L.ptr(g), g.numel(), L.ptr(ws), float(scale), float(max_norm), L.ptr(out2), L.stream()),
            "s2t_grad_norm_clip")
    return out2


def adam_step(p, g, m, v, shadow, mult2, lr, beta1, beta2, eps, wd, step):
    L.check(_lib().s2t_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(shadow), p.numel(), L.ptr(mult2), float(lr),
                                 float(beta1), float(beta2), float(eps), float(wd), int(step), L.stream()), "s2t_adam_step")


def cast(src, dst):
    assert src.numel() == dst.numel()
    L.check(_lib().s2t_cast(L.dt(src), L.dt(dst), L.ptr(src), L.ptr(dst), src.numel(), L.stream()), "s2t_cast")
    return dst


def scale_by_device_scalar(x, scalar):
    d = x if x.is_contiguous() else x._base           # row-padded views: scale the whole backing buffer
    assert d is not None and d.is_contiguous()
    L.check(_lib().s2t_scale_by_device_scalar(L.dt(d), L.ptr(d), d.numel(), L.ptr(scalar), L.stream()), "s2t_scale")
    return x


def host_ctc_uer(pred_cpu, in_len_cpu, targets_cpu, tgt_len_cpu, blank):
    """HOST tensors (int32 pred [B,T], int64 others) -> (errors, total)."""
    B, T = pred_cpu.shape
    e, n = ctypes.c_double(0), ctypes.c_double(0)
    pred_cpu = pred_cpu.contiguous(); targets_cpu = targets_cpu.contiguous()
    L.check(_lib().s2t_host_ctc_uer(pred_cpu.data_ptr(), in_len_cpu.contiguous().data_ptr(), B, T, targets_cpu.data_ptr(),
                                    tgt_len_cpu.contiguous().data_ptr(), targets_cpu.shape[1], blank,
                                    ctypes.addressof(e), ctypes.addressof(n)), "s2t_host_ctc_uer")
    return e.value, n.value


OPTION_EPOCH = 0          # bumps with every set_option: sizes that depend on the kernel routes (the engine's cached workspace sizes) key on it


def set_option(key, value):
    """kernel-route option of the library (include/s2t_hip.h, s2t_set_option); returns the previous value"""
    global OPTION_EPOCH
    old = _lib().s2t_set_option(key.encode(), int(value))
    if old == -22:
        raise ValueError("unknown libs2t_hip option %r (or a value out of its range)" % key)
    OPTION_EPOCH += 1
    return old


def prof_enable(on):
    _lib().s2t_prof_enable(int(on))


def prof_reset():
    _lib().s2t_prof_reset()


def prof_read(family):
    ms, fl, by = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    n = ctypes.c_longlong(0)
    _lib().s2t_prof_read(family.encode(), ctypes.addressof(ms), ctypes.addressof(n), ctypes.addressof(fl), ctypes.addressof(by))
    return dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)


# ------------------------------------------------------------------ ConvAttention2D pieces (csrc/attn2d.hip)
def a2d_chan_stats(z, C, Cg, prescale=None, dy=None, bn=None):
    """z [M, ld] channels-last; returns double sums in groups of Cg ([Cg | Cg] each).  dy + bn=(mean, rstd, scale, shift): backward sums."""
    M = z.shape[0]
    sums = torch.zeros((2 * C,), dtype=torch.float64, device=z.device)
    mean, rstd, scale, shift = bn if bn is not None else (None, None, None, None)
    L.check(_lib().s2t_a2d_chan_stats(L.dt(z), L.ptr(z), L.ptr(dy), L.ptr(prescale), L.ptr(mean), L.ptr(rstd), L.ptr(scale), L.ptr(shift),
                                      L.ptr(sums), M, C, z.stride(0), dy.stride(0) if dy is not None else 0, Cg,
                                      0 if dy is None else 1, L.stream()), "s2t_a2d_chan_stats")
    return sums


def a2d_bn_act(z, C, scale, shift, prescale=None, res=None, out=None):
    M = z.shape[0]
    if out is None:
        out = torch.zeros_like(z) if res is None else torch.empty_like(res)
    assert res is None or res.stride(0) == out.stride(0)
    L.check(_lib().s2t_a2d_bn_act(L.dt(z), L.ptr(z), L.ptr(prescale), L.ptr(scale), L.ptr(shift), L.ptr(res), L.ptr(out), M, C,
                                  z.stride(0), out.stride(0), L.stream()), "s2t_a2d_bn_act")
    return out


def a2d_bn_bwd(dy, z, C, Cg, bn, sums, count, training, prescale=None):
    mean, rstd, scale, shift = bn
    dz = torch.zeros_like(z)
    L.check(_lib().s2t_a2d_bn_bwd(L.dt(z), L.ptr(dy), L.ptr(z), L.ptr(prescale), L.ptr(mean), L.ptr(rstd), L.ptr(scale), L.ptr(shift),
                                  L.ptr(sums), L.ptr(dz), z.shape[0], C, dy.stride(0), z.stride(0), Cg, float(count), int(training),
                                  L.stream()), "s2t_a2d_bn_bwd")
    return dz


def a2d_param_grads(sums, dgamma, dbeta):
    L.check(_lib().s2t_a2d_param_grads(L.ptr(sums), L.ptr(dgamma), L.ptr(dbeta), dgamma.numel(), L.stream()), "s2t_a2d_param_grads")


def a2d_pack_w(w, rows, CP, dtype, mode):
    """mode 0 / 1: packed operand [rows, 9*CP] of the forward / data-gradient gathered GEMM from nn.Conv2d weights [Co,Ci,3,3]."""
    Co, Ci = w.shape[0], w.shape[1]
    dst = torch.zeros((rows, 9 * CP), dtype=dtype, device=w.device)
    L.check(_lib().s2t_a2d_pack_w(L.dt(dst), L.ptr(w), L.ptr(dst), 0, Co, Ci, CP, dst.stride(0), mode, L.stream()), "s2t_a2d_pack_w")
    return dst


def a2d_unpack_wgrad(gp, grad, CP):
    """grad[Co,Ci,3,3] += gp[Co, 9*CP] (f32)"""
    Co, Ci = grad.shape[0], grad.shape[1]
    assert gp.dtype == torch.float32 and grad.dtype == torch.float32
    L.check(_lib().s2t_a2d_pack_w(L.F32, L.ptr(gp), 0, L.ptr(grad), Co, Ci, CP, gp.stride(0), 2, L.stream()), "s2t_a2d_pack_w")


def a2d_time_fwd(qkv, cat, B, T, F, p_drop=0.0, seed=0):
    lse = torch.empty((B * 4, T), dtype=torch.float32, device=qkv.device)
    L.check(_lib().s2t_a2d_time_fwd(L.dt(qkv), L.ptr(qkv), L.ptr(cat), L.ptr(lse), B, T, F, float(p_drop), int(seed), L.stream()),
            "s2t_a2d_time_fwd")
    return lse


def a2d_time_bwd(qkv, cat, dcat, lse, dqkv, B, T, F, p_drop=0.0, seed=0):
    delta = torch.empty_like(lse)
    L.check(_lib().s2t_a2d_time_bwd(L.dt(qkv), L.ptr(qkv), L.ptr(cat), L.ptr(dcat), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), B, T, F,
                                    float(p_drop), int(seed), L.stream()), "s2t_a2d_time_bwd")


def a2d_freq_fwd(qkv, cat, B, T, F, p_drop=0.0, seed=0):
    A = torch.empty((B * 4, F, F), dtype=torch.float32, device=qkv.device)
    L.check(_lib().s2t_a2d_freq_fwd(L.dt(qkv), L.ptr(qkv), L.ptr(cat), L.ptr(A), B, T, F, float(p_drop), int(seed), L.stream()),
            "s2t_a2d_freq_fwd")
    return A


def a2d_freq_bwd(qkv, dcat, A, dqkv, B, T, F, p_drop=0.0, seed=0):
    L.check(_lib().s2t_a2d_freq_bwd(L.dt(qkv), L.ptr(qkv), L.ptr(dcat), L.ptr(A), L.ptr(dqkv), B, T, F, float(p_drop), int(seed),
                                    L.stream()), "s2t_a2d_freq_bwd")


def a2d_conv_wgrad(dy, x, grad, B, T, F):
    """grad[Co,Ci,3,3] += conv-weight gradient from dy [M, >=Co] and x [M, Ci] (pixel rows (t, b, f)); False when the shape is not built"""
    Co, Ci = grad.shape[0], grad.shape[1]
    assert grad.dtype == torch.float32 and grad.is_contiguous() and dy.dtype == x.dtype
    ws = torch.empty((512, max(Co, 16) * Ci * 9), dtype=torch.float32, device=x.device)      # S2T_A2D_WGRAD_GROUPS partial sums
    rc = _lib().s2t_a2d_conv_wgrad(L.dt(x), L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), L.ptr(grad), L.ptr(ws), Co, Ci, B, T, F,
                                   L.stream())
    if rc == -95:
        return False
    L.check(rc, "s2t_a2d_conv_wgrad")
    return True


def a2d_planes(chl, planes, ch0, B, T, F, to_planes):
    """chl [M, ld] channels-last pixel rows <-> planes [G, T, B, 128] (head-major, F columns of 32 used per head)"""
    G = planes.shape[0]
    assert planes.is_contiguous() and planes.shape[1:] == (T, B, 128) and chl.stride(1) == 1 and planes.dtype == chl.dtype
    L.check(_lib().s2t_a2d_planes(L.dt(chl), L.ptr(chl), L.ptr(planes), G, ch0, chl.stride(0), B, T, F, 0 if to_planes else 1, L.stream()),
            "s2t_a2d_planes")
    return planes if to_planes else chl
