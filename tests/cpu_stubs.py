"""CPU stand-ins for the HIP side, for tests of HOST logic that must run where there is no GPU (test infrastructure only).

The product has no CPU path: these stand in for `libs2t_hip.so` entry points (a handful of `kernels.*` wrappers) and for
`engine.S2TEngine` in tests whose subject is the Python boundary around them -- the reference's own Trainer / SequenceGenerator
driving a plug-in model (tests/test_reference_trainer_cpu.py, tests/golden/make_trainer_fixture.py), in the pattern of the stub
engine of tests/test_trainer_dp_cpu.py.  Nothing under fbk_fairseq_st_amd/ imports this file.
"""
import contextlib

import torch

from oracle import s2t_ref


# ------------------------------------------------------------------ kernels.* stand-ins (same signatures, torch CPU arithmetic)
def _lsce(logits, target, eps, pad, want_grad=True, grad_scale=1.0):
    with torch.enable_grad():                                  # called from inside an autograd.Function.forward
        x = logits.detach().float().clone().requires_grad_(True)
        loss, nll = s2t_ref.label_smoothed_nll(x, target, eps, pad)
        (g,) = torch.autograd.grad(loss, x)
    return torch.stack([loss.detach(), nll.detach()]).float(), (g * grad_scale).to(logits.dtype) if want_grad else None


def _scale_by_device_scalar(x, scalar):
    x.mul_(scalar.reshape(()).to(x.dtype))
    return x


def _grad_norm_clip(g, scale, max_norm, ws, out2, divisor=None):
    """out2[0] = |scale * g|, out2[1] = scale * min(1, max_norm / (norm + 1e-6)) (0 disables clipping) -- csrc/norm_optim.hip;
    divisor (f64 scalar tensor): the gradients are also divided by max(divisor, 1), folded into `scale` (s2t_grad_norm_clip_div)"""
    if divisor is not None:
        scale = scale / max(float(divisor), 1.0)
    norm = (g.double() * scale).norm()
    coef = 1.0 if max_norm <= 0 else min(1.0, float(max_norm) / (float(norm) + 1e-6))
    out2[0] = float(norm)
    out2[1] = scale * coef
    return out2


def _adam_step(p, g, m, v, shadow, mult2, lr, beta1, beta2, eps, wd, step):
    gg = g * mult2[1]
    pn, mn, vn = s2t_ref.adam_step(p, gg, m, v, step, lr, beta1, beta2, eps, wd)
    p.copy_(pn); m.copy_(mn); v.copy_(vn)
    if shadow is not None:
        shadow.copy_(p.to(shadow.dtype))


def _cast(src, dst):
    dst.copy_(src.to(dst.dtype))
    return dst


def _log_softmax(logits, temperature=1.0):
    return torch.log_softmax(logits.float() / float(temperature), dim=-1)


STANDINS = {"lsce": _lsce, "scale_by_device_scalar": _scale_by_device_scalar, "grad_norm_clip": _grad_norm_clip,
            "adam_step": _adam_step, "cast": _cast, "log_softmax": _log_softmax}


@contextlib.contextmanager
def cpu_kernels():
    """swap the listed kernels.* wrappers for their stand-ins for the duration of the block"""
    from fbk_fairseq_st_amd import kernels as K
    saved = {k: getattr(K, k) for k in STANDINS}
    for k, f in STANDINS.items():
        setattr(K, k, f)
    try:
        yield
    finally:
        for k, f in saved.items():
            setattr(K, k, f)


# ------------------------------------------------------------------ a toy differentiable engine (training-step plumbing)
def toy_forward(W, src_tokens, prev_tokens):
    """The function the toy engine computes, on a dict of reference-named... arena-named tensors: every frame of the encoder output
    is fc3.weight[:, :F] @ mean_t(src); a decoder row is embed[prev] + mean_t(encoder_out); logits = row @ output_projection^T.
    Three parameters receive gradients; everything else in the arena stays untouched (zero gradient)."""
    F_ = src_tokens.shape[2]
    e = src_tokens.float().mean(1) @ W["encoder.fc3.weight"][:, :F_].t()                    # [B, D]
    T4 = ((src_tokens.shape[1] + 1) // 2 + 1) // 2
    enc_out = e.unsqueeze(0).expand(T4, -1, -1)
    h = W["decoder.embed_tokens.weight"][prev_tokens.t()] + enc_out.mean(0)                  # [L, B, D]
    return enc_out, h @ W["decoder.output_projection.weight"].t()                            # [T4,B,D], [L,B,V]


class ToyEngine:
    """encoder_forward / encoder_backward / decoder_forward / decoder_backward of engine.S2TEngine for toy_forward, writing
    gradients into the arena and reporting finished parameter groups like the real one"""

    def __init__(self, model):
        self.model, self.A = model, model.arena
        self.on_grads_ready = None
        self.calls = []

    def reset_wgrad(self):
        pass

    def flush_wgrad(self):
        pass

    def _ready(self, prefix):
        if self.on_grads_ready is not None:
            self.on_grads_ready(prefix)

    def encoder_forward(self, src_tokens, src_lengths, training, seed=0, return_all_hiddens=False, keep=None):
        self.calls.append(("encoder_forward", bool(training), int(seed)))
        lens = [((int(l) + 1) // 2 + 1) // 2 for l in src_lengths.tolist()]
        T4 = ((src_tokens.shape[1] + 1) // 2 + 1) // 2
        with torch.enable_grad():
            w = self.A.p("encoder.fc3.weight").detach().clone().requires_grad_(True)
            F_ = src_tokens.shape[2]
            e = src_tokens.float().mean(1) @ w[:, :F_].t()
            out = e.unsqueeze(0).expand(T4, -1, -1).contiguous()
        res = dict(out=out.detach(), ctc_out=None, ctc_lengths=None, ctc_lengths_host=None, pred=None, states=None,
                   lengths=torch.tensor(lens, dtype=torch.int64), lengths_host=lens, klen=None, ctc_klen=None)
        return res, dict(graph=(out, w), ctc=None, state_layers=[])

    def encoder_backward(self, ctx, d_out, d_ctc_out=None, d_states=None):
        out, w = ctx["graph"]
        out.backward(d_out.reshape(out.shape))
        self.A.g("encoder.fc3.weight").add_(w.grad)
        self._ready("encoder.")

    def decoder_forward(self, prev_tokens, enc_out, enc_klen32, training, seed=0, pfx="decoder.", keep=None, attn_layer=None, attn_heads=None):
        self.calls.append(("decoder_forward", bool(training), int(seed)))
        B, L = prev_tokens.shape
        with torch.enable_grad():
            E = self.A.p(pfx + "embed_tokens.weight").detach().clone().requires_grad_(True)
            Wo = self.A.p(pfx + "output_projection.weight").detach().clone().requires_grad_(True)
            eo = enc_out.detach().clone().requires_grad_(True)
            h = E[prev_tokens.t()] + eo.mean(0)
            logits = (h @ Wo.t()).reshape(L * B, -1)
        return logits.detach(), dict(graph=(logits, E, Wo, eo), Ts=enc_out.shape[0], B=B, pfx=pfx)

    def decoder_backward(self, ctx, dlogits, denc=None):
        logits, E, Wo, eo = ctx["graph"]
        logits.backward(dlogits.reshape(logits.shape))
        pfx = ctx["pfx"]
        self.A.g(pfx + "embed_tokens.weight").add_(E.grad)
        self.A.g(pfx + "output_projection.weight").add_(Wo.grad)
        self._ready(pfx)
        g = eo.grad.reshape(ctx["Ts"] * ctx["B"], -1)
        return g if denc is None else denc.add_(g)


# ------------------------------------------------------------------ the oracle as an engine (generation plumbing)
class OracleEngine:
    """encoder_forward / decoder_begin / decoder_step / decoder_reorder of engine.S2TEngine answered by oracle/s2t_ref.py on the
    host: lets the reference's SequenceGenerator run over the plug-in's encoder / incremental decoder objects without a GPU.
    The decoder state is the token history per hypothesis (re-scored in full every step: a checker, not a fast path)."""

    def __init__(self, model, W, cfg):
        self.model, self.W, self.cfg = model, W, cfg

    def encoder_forward(self, src_tokens, src_lengths, training, seed=0, return_all_hiddens=False, keep=None):
        lens = src_lengths if torch.is_tensor(src_lengths) else torch.tensor(src_lengths)
        enc, _ = s2t_ref.encoder_forward(self.W, self.cfg, src_tokens, lens, training=False)
        L = [int(v) for v in enc.src_lengths]
        T = enc.encoder_out.shape[0]
        klen = torch.tensor(L, dtype=torch.int32) if min(L) < T else None
        res = dict(out=enc.encoder_out.detach(), ctc_out=None, ctc_lengths=None, ctc_lengths_host=None, pred=None, states=None,
                   lengths=torch.tensor(L, dtype=torch.int64), lengths_host=L, klen=klen, ctc_klen=None)
        return res, dict(ctc=None, state_layers=[])

    def decoder_begin(self, enc_out, enc_klen32, max_steps, pfx="decoder."):
        return dict(pfx=pfx, enc=enc_out, klen=enc_klen32, toks=None, steps=0, max_steps=max_steps, N=enc_out.shape[1])

    def decoder_reorder(self, st, order, encoder_side=False):
        if st["toks"] is not None:
            st["toks"] = st["toks"].index_select(0, order)
        if encoder_side:
            st["enc"] = st["enc"].index_select(1, order)
            if st["klen"] is not None:
                st["klen"] = st["klen"].index_select(0, order)
        st["N"] = int(order.numel())

    def decoder_step(self, st, last_tokens):
        t = last_tokens.view(-1, 1)
        st["toks"] = t if st["toks"] is None else torch.cat([st["toks"], t], 1)
        enc = st["enc"]
        mask = None
        if st["klen"] is not None:
            mask = torch.arange(enc.shape[0])[None, :] >= st["klen"][:, None].long()
        logits = s2t_ref.decoder_forward(self.W, self.cfg, st["toks"], enc, mask, pfx=st["pfx"])
        st["steps"] += 1
        return logits[:, -1, :].detach()


# ------------------------------------------------------------------ the oracle as a TRAINING engine (whole-CLI plumbing)
def cfg_of(hp):
    """oracle cfg dict of an engine.HParams"""
    return s2t_ref.default_cfg(D=hp.D, heads=hp.heads, ffn=hp.ffn, enc_layers=hp.enc_layers, dec_layers=hp.dec_layers,
                               ctc_layer=hp.ctc_layer, act=hp.act, pad=hp.pad, strategy=("avg", "weighted", "softmax")[hp.ctc_strategy],
                               conv_ch=hp.conv_ch, feat=hp.feat, no_scale_embedding=hp.no_scale_embedding, ln_eps=hp.ln_eps,
                               bn_eps=hp.bn_eps, bn_momentum=hp.bn_momentum, attn_2d=hp.attn_2d,
                               distance_penalty=bool(hp.distance_penalty))


class OracleTrainEngine(OracleEngine):
    """engine.S2TEngine's training AND generation entry points answered by oracle/s2t_ref.py under torch autograd on the host,
    reading the weights from the arena and adding the gradients into it: with it the reference's unchanged `train.main` /
    `generate.main` run end to end over the plug-in where there is no GPU (tests/golden/make_cli_fixture.py).  Dropout is the
    identity, as everywhere in the oracle; the constructor has S2TEngine's signature so that
    `conv_transformer.S2TEngine = OracleTrainEngine` is the whole installation."""

    def __init__(self, hp, arena):
        self.hp, self.A, self.cfg = hp, arena, cfg_of(hp)
        self.on_grads_ready = None
        self.bn_buffers = {}
        self.wgrad_flush_layers = None
        self.calls = []

    # -- weights: one autograd leaf per arena tensor, handed to the oracle under the reference's names (views of the leaves)
    def _weights(self, prefixes):
        from fbk_fairseq_st_amd.conv_transformer import fused_to_reference
        leaves = {n: self.A.p(n).detach().clone().requires_grad_(True) for n in self.A.slices if n.startswith(prefixes)}
        W = {}
        for k, v in leaves.items():
            if ".self_attn.qkv." in k:
                D = v.shape[0] // 3
                for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    W[k.replace("qkv", n)] = v[i * D:(i + 1) * D]
            elif ".encoder_attn.kv." in k:
                D = v.shape[0] // 2
                for i, n in enumerate(("k_proj", "v_proj")):
                    W[k.replace("kv", n)] = v[i * D:(i + 1) * D]
            else:
                W[k] = v
        for k, b in self.bn_buffers.items():
            if k.startswith(prefixes) and not k.endswith("num_batches_tracked"):
                W[k] = b.detach().clone()
        assert fused_to_reference is not None
        return leaves, W

    def _harvest(self, leaves, prefix):
        for n, v in leaves.items():
            if v.grad is not None:
                self.A.g(n).add_(v.grad)
        self._ready(prefix)

    def _ready(self, prefix):
        if self.on_grads_ready is not None:
            self.on_grads_ready(prefix)

    def reset_wgrad(self):
        pass

    def flush_wgrad(self):
        pass

    def W(self, name):
        return self.A.p(name)

    P = W

    # -- encoder
    def encoder_forward(self, src_tokens, src_lengths, training, seed=0, return_all_hiddens=False, keep=None):
        self.calls.append(("encoder_forward", bool(training), int(seed)))
        lens = src_lengths if torch.is_tensor(src_lengths) else torch.tensor(src_lengths)
        cfg = dict(self.cfg, enc_keep=keep)
        with torch.enable_grad():
            leaves, W = self._weights(("encoder.",))
            enc, stats = s2t_ref.encoder_forward(W, cfg, src_tokens.float(), lens.long(), training=training)
        if training:                                   # BatchNorm running statistics (nn.BatchNorm2d, momentum 0.1)
            for k, v in stats.items():
                self.bn_buffers[k].copy_(v.detach())
            for k, b in self.bn_buffers.items():
                if k.startswith("encoder.") and k.endswith("num_batches_tracked"):
                    b.add_(1)
        L = [int(v) for v in enc.src_lengths]
        T = enc.encoder_out.shape[0]
        res = dict(out=enc.encoder_out.detach(), ctc_out=None, ctc_lengths=None, ctc_lengths_host=None, pred=None,
                   states=[s.detach() for s in enc.encoder_states] if return_all_hiddens else None,
                   lengths=torch.tensor(L, dtype=torch.int64), lengths_host=L,
                   klen=torch.tensor(L, dtype=torch.int32) if min(L) < T else None, ctc_klen=None)
        if enc.ctc_out is not None:
            T4 = enc.ctc_out.shape[0]
            L4 = [((int(l) + 1) // 2 + 1) // 2 for l in lens.tolist()]
            res.update(ctc_out=enc.ctc_out.detach(), ctc_lengths=torch.tensor(L4, dtype=torch.int64), ctc_lengths_host=L4,
                       ctc_klen=torch.tensor(L4, dtype=torch.int32) if min(L4) < T4 else None, pred=enc.ctc_pred.to(torch.int32),
                       pred_host=enc.ctc_pred.to(torch.int32))
        layers = [l for l in range(self.hp.enc_layers) if keep is None or keep[l]]
        return res, dict(graph=(enc, leaves), ctc=enc.ctc_out is not None or None, state_layers=layers)

    def encoder_backward(self, ctx, d_out, d_ctc_out=None, d_states=None):
        enc, leaves = ctx["graph"]
        outs, grads = [enc.encoder_out], [d_out.reshape(enc.encoder_out.shape)]
        if d_ctc_out is not None:
            outs.append(enc.ctc_out)
            grads.append(d_ctc_out.reshape(enc.ctc_out.shape))
        for l, g in (d_states or {}).items():
            s = enc.encoder_states[ctx["state_layers"].index(l)]
            outs.append(s)
            grads.append(g.reshape(s.shape))
        torch.autograd.backward(outs, grads)
        self._harvest(leaves, "encoder.")

    # -- decoder
    def decoder_forward(self, prev_tokens, enc_out, enc_klen32, training, seed=0, pfx="decoder.", keep=None, attn_layer=None, attn_heads=None):
        self.calls.append(("decoder_forward", bool(training), int(seed)))
        B, L = prev_tokens.shape
        mask = None
        if enc_klen32 is not None:
            mask = torch.arange(enc_out.shape[0])[None, :] >= enc_klen32[:, None].long()
        with torch.enable_grad():
            leaves, W = self._weights((pfx,))
            eo = enc_out.detach().clone().requires_grad_(True)
            logits = s2t_ref.decoder_forward(W, dict(self.cfg, dec_keep=keep), prev_tokens, eo, mask, pfx=pfx, attn_layer=attn_layer,
                                             attn_heads=attn_heads)   # (B, L, V)
            attn = None
            if attn_layer is not None:
                logits, attn = logits
            tm = logits.transpose(0, 1).reshape(L * B, -1)
        ctx = dict(graph=(tm, leaves, eo), Ts=enc_out.shape[0], B=B, pfx=pfx)
        if attn is not None:
            ctx["attn"] = attn.detach().float()
        return tm.detach(), ctx

    def decoder_backward(self, ctx, dlogits, denc=None):
        tm, leaves, eo = ctx["graph"]
        tm.backward(dlogits.reshape(tm.shape))
        self._harvest(leaves, ctx["pfx"])
        g = eo.grad.reshape(ctx["Ts"] * ctx["B"], -1)
        return g if denc is None else denc.add_(g)

    def linear_bwd(self, dy2, x2, stem):
        """criterion-owned head (criterions._LinearFn.backward): dW, db into the arena, returns dx"""
        self.A.g(stem + ".weight").add_(dy2.float().t() @ x2.float())
        self.A.g(stem + ".bias").add_(dy2.float().sum(0))
        return dy2.float() @ self.A.p(stem + ".weight")

    # -- generation: the weights are read from the arena at decoder_begin
    def decoder_begin(self, enc_out, enc_klen32, max_steps, pfx="decoder."):
        with torch.no_grad():
            _, self.W_gen = self._weights((pfx,))
        return dict(pfx=pfx, enc=enc_out, klen=enc_klen32, toks=None, steps=0, max_steps=max_steps, N=enc_out.shape[1])

    def decoder_step(self, st, last_tokens):
        t = last_tokens.view(-1, 1)
        st["toks"] = t if st["toks"] is None else torch.cat([st["toks"], t], 1)
        enc = st["enc"]
        mask = None
        if st["klen"] is not None:
            mask = torch.arange(enc.shape[0])[None, :] >= st["klen"][:, None].long()
        with torch.no_grad():
            logits = s2t_ref.decoder_forward(self.W_gen, self.cfg, st["toks"], enc, mask, pfx=st["pfx"], attn_layer=st.get("attn_layer"),
                                             attn_heads=st.get("attn_heads"))
            if st.get("attn_layer") is not None:
                logits, attn = logits
                st["attn"] = attn[:, -1:, :].float()
        st["steps"] += 1
        return logits[:, -1, :].detach()


class _CTCFnCPU(torch.autograd.Function):
    """criterions._CTCFn without its streams: the oracle's CTC loss (float64 recursion) and its gradient"""

    @staticmethod
    def forward(ctx, logits, targets, tgt_len, in_len32, blank, lse=None):
        with torch.enable_grad():
            x = logits.detach().float().clone().requires_grad_(True)
            loss = s2t_ref.ctc_loss_sum(x, targets, in_len32.long(), tgt_len, blank)
            (g,) = torch.autograd.grad(loss, x)
        ctx.g = g
        return loss.detach()

    @staticmethod
    def backward(ctx, g):
        return ctx.g * g, None, None, None, None, None

    @staticmethod
    def join():
        pass


def _ctc_argmax(x_ctc, want_lse=False):
    from oracle import int_ref
    import numpy as np
    lp = torch.log_softmax(x_ctc.float(), dim=-1).transpose(0, 1)
    pred = torch.from_numpy(int_ref.argmax_first_np(lp.detach().numpy()).astype(np.int32))
    return (pred, None, None) if want_lse else (pred, None)


def _matmul_linear(x2, w, bias=None, out=None, **kw):
    y = x2.float() @ w.float().t()
    return y + bias if bias is not None else y


@contextlib.contextmanager
def oracle_engine():
    """everything the reference's CLI needs to run over the plug-in on the host: the kernels.* stand-ins, the oracle as the
    engine class the model instantiates, the stream-free CTC bridge, and no `cuda only` refusal"""
    from fbk_fairseq_st_amd import conv_transformer as CT, criterions as CR, kernels as K
    saved = (CT.S2TEngine, CR._CTCFn, CT.ConvolutionalTransformerModel._ensure_engine, K.ctc_argmax, K.gemm, K.alloc_rows)

    def ensure(self, device):
        if self.engine is None:
            self.materialize(device, self.compute_dtype)

    CT.S2TEngine, CR._CTCFn, CT.ConvolutionalTransformerModel._ensure_engine = OracleTrainEngine, _CTCFnCPU, ensure
    K.ctc_argmax, K.gemm = _ctc_argmax, _matmul_linear
    K.alloc_rows = lambda lead, n, dtype, device: None
    try:
        with cpu_kernels():
            yield
    finally:
        CT.S2TEngine, CR._CTCFn, CT.ConvolutionalTransformerModel._ensure_engine, K.ctc_argmax, K.gemm, K.alloc_rows = saved
