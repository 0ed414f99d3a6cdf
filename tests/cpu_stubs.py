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


def _grad_norm_clip(g, scale, max_norm, ws, out2):
    """out2[0] = |scale * g|, out2[1] = scale * min(1, max_norm / (norm + 1e-6)) (0 disables clipping) -- csrc/norm_optim.hip"""
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

    def decoder_forward(self, prev_tokens, enc_out, enc_klen32, training, seed=0, pfx="decoder.", keep=None):
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
