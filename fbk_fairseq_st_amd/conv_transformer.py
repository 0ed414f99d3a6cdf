"""`conv_transformer` (FBK S-Transformer) behind the reference's model interface, running on the HIP engine.

Mirrors examples/speech_recognition/models/conv_transformer.py: same registry names
(`--arch conv_transformer{,_big,_big2,_giant}`), same flags (add_args :47-72), same
encoder/decoder call signatures and output tuples, same state-dict key names.
On top of the reference's presets it registers the build-defined `s2t_transformer{,_xs,_s,_m,_l}`
presets that BASELINE.json names.

`--distance-penalty log` (LocalAttention), the ConvAttention2D front end (N3) and LayerDrop
(`--encoder-layerdrop` / `--decoder-layerdrop`) are supported.  Out of scope here:
`--distance-penalty gauss`, learned positions, adaptive softmax.
"""
import math
import weakref

import torch
import torch.nn as nn

from . import kernels as K
from .arena import ParamArena
from .engine import HParams, S2TEngine
from .registry import (CTCAwareEncoderOut, EncoderOut, FairseqEncoder, FairseqEncoderDecoderModel,
                       FairseqIncrementalDecoder, register_model, register_model_architecture)

STRATEGIES = {"avg": 0, "weighted": 1, "softmax": 2}


# ------------------------------------------------------------------ state-dict name mapping
def reference_to_fused(sd, prefix_filter=None):
    """reference keys (q_proj/k_proj/v_proj separate) -> arena keys (qkv / kv fused)."""
    out = {}
    done = set()
    for k, v in sd.items():
        if k in done:
            continue
        for kind in ("weight", "bias"):
            if k.endswith("self_attn.q_proj." + kind):
                base = k[: -len("q_proj." + kind)]
                out[base + "qkv." + kind] = torch.cat([sd[base + "q_proj." + kind], sd[base + "k_proj." + kind], sd[base + "v_proj." + kind]], 0)
                done.update(base + n + "." + kind for n in ("q_proj", "k_proj", "v_proj"))
                break
            if k.endswith("encoder_attn.k_proj." + kind):
                base = k[: -len("k_proj." + kind)]
                out[base + "kv." + kind] = torch.cat([sd[base + "k_proj." + kind], sd[base + "v_proj." + kind]], 0)
                done.update(base + n + "." + kind for n in ("k_proj", "v_proj"))
                break
        else:
            if ("self_attn.k_proj" in k or "self_attn.v_proj" in k or "encoder_attn.v_proj" in k):
                continue
            out[k] = v
    return out


def fused_to_reference(sd):
    out = {}
    for k, v in sd.items():
        if ".self_attn.qkv." in k:
            D = v.shape[0] // 3
            for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[k.replace("qkv", n)] = v[i * D:(i + 1) * D].clone()
        elif ".encoder_attn.kv." in k:
            D = v.shape[0] // 2
            for i, n in enumerate(("k_proj", "v_proj")):
                out[k.replace("kv", n)] = v[i * D:(i + 1) * D].clone()
        else:
            out[k] = v
    return out


def reference_slot(name):
    """reference parameter name -> (arena name, block index or None, blocks): where that parameter's rows live in the fused
    layout (q|k|v of a self-attention, k|v of an encoder attention; LocalAttention's in_proj IS the fused tensor)."""
    for kind in ("weight", "bias"):
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            if name.endswith(".self_attn.%s.%s" % (n, kind)):
                return name[: -len("%s.%s" % (n, kind))] + "qkv." + kind, i, 3
        for i, n in enumerate(("k_proj", "v_proj")):
            if name.endswith(".encoder_attn.%s.%s" % (n, kind)):
                return name[: -len("%s.%s" % (n, kind))] + "kv." + kind, i, 2
        if name.endswith(".self_attn.in_proj_" + kind):
            return name[: -len("in_proj_" + kind)] + "qkv." + kind, None, 1
    if name.startswith("ctc_aware_model."):                      # criterion-owned head (ctc_multi_loss.py:14-46)
        return "criterion." + name, None, 1
    return name, None, 1


# ------------------------------------------------------------------ autograd bridges
def anchor_zero_grad(anchor):
    z = getattr(anchor, "_s2t_zero", None)
    if z is None or z.device != anchor.device:
        z = anchor._s2t_zero = torch.zeros_like(anchor.data)
    return z


def unwrap_model(model):
    """the arena-homed model behind a data-parallel wrapper (fairseq/models/distributed_fairseq_model.py:88-100 forwards attribute
    READS to the wrapped module; attribute writes would land on the wrapper)"""
    while not hasattr(type(model), "_ensure_engine") and isinstance(getattr(model, "module", None), nn.Module):
        model = model.module
    return model


class _EncoderFn(torch.autograd.Function):
    """Connects the engine's encoder to torch autograd: outputs (encoder_out, ctc_out, state_k)."""

    @staticmethod
    def forward(ctx, anchor, enc, src_tokens, src_lengths, training, seed, want_state, keep=None):
        out, ectx = enc.engine.encoder_forward(src_tokens, src_lengths, training, seed, return_all_hiddens=want_state is not None,
                                               keep=keep)
        ctx.enc, ctx.ectx, ctx.want_state, ctx.anchor = enc, ectx, want_state, anchor
        enc._last = out
        eo = out["out"]
        co = out["ctc_out"] if out["ctc_out"] is not None else eo.new_zeros(1)
        st = out["states"][want_state] if want_state is not None else eo.new_zeros(1)
        ctx.mark_non_differentiable(*[t for t, keep in ((co, out["ctc_out"] is not None), (st, want_state is not None)) if not keep])
        return eo, co, st

    @staticmethod
    def backward(ctx, d_out, d_ctc, d_state):
        eng = ctx.enc.engine
        has_ctc = ctx.ectx["ctc"] is not None
        # `states` holds one entry per layer that RAN (LayerDrop): entry -> layer index
        ds = ({ctx.ectx["state_layers"][ctx.want_state]: d_state.contiguous()}
              if (ctx.want_state is not None and d_state is not None) else None)
        eng.encoder_backward(ctx.ectx, d_out, d_ctc if (has_ctc and d_ctc is not None) else None, ds)
        # The encoder's backward is the LAST piece of a backward pass and ends with the flush of the deferred weight gradients: every
        # gradient of the arena is final here.  A data-parallel wrapper that reduces `p.grad` in place from a parameter hook -- the
        # reference's LegacyDistributedDataParallel (`--ddp-backend no_c10d`, legacy_distributed_data_parallel.py:173-180) -- never
        # sees an autograd gradient on this path (the kernels write the arena), so the anchor, the one parameter autograd does know,
        # is handed a zero gradient when somebody hooked it: the hook queues the wrapper's reduction for the end of this pass.
        ga = anchor_zero_grad(ctx.anchor) if ctx.anchor._backward_hooks else None
        return (ga,) + (None,) * 7


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, dec, prev_tokens, enc_out, enc_klen, training, seed, keep=None, attn_layer=None, attn_heads=None):
        kw = {} if attn_layer is None else dict(attn_layer=attn_layer, attn_heads=attn_heads)
        logits, dctx = dec.engine.decoder_forward(prev_tokens, enc_out.contiguous(), enc_klen, training, seed, pfx=dec.pfx, keep=keep, **kw)
        ctx.dec, ctx.dctx = dec, dctx
        attn = dctx.pop("attn", None)
        if attn is None:
            attn = logits.new_zeros((0,), dtype=torch.float32)
        ctx.mark_non_differentiable(attn)              # the reference's attention weights are never differentiated on this path
        return logits, attn

    @staticmethod
    def backward(ctx, dlogits, _dattn=None):
        denc = ctx.dec.engine.decoder_backward(ctx.dctx, dlogits if dlogits.stride(-1) == 1 else dlogits.contiguous())
        c = ctx.dctx
        return None, None, None, denc.view(c["Ts"], c["B"], -1), None, None, None, None, None, None


class _DualDecoderFn(torch.autograd.Function):
    """Both decoders of the dual-decoder model over one encoder output; the two encoder-output gradients are
    accumulated by the kernels into one buffer (no autograd add)."""

    @staticmethod
    def forward(ctx, anchor, model, prev_tokens, aux_prev_tokens, enc_out, enc_klen, training, seed):
        eng = model.engine
        eo = enc_out.contiguous()
        l1, c1 = eng.decoder_forward(prev_tokens, eo, enc_klen, training, seed, pfx="decoder.")
        l2, c2 = eng.decoder_forward(aux_prev_tokens, eo, enc_klen, training, seed + 1, pfx="auxiliary_decoder.")
        ctx.model, ctx.c1, ctx.c2 = model, c1, c2
        return l1, l2

    @staticmethod
    def backward(ctx, d1, d2):
        eng = ctx.model.engine
        fix = lambda d: d if d.stride(-1) == 1 else d.contiguous()
        denc = eng.decoder_backward(ctx.c2, fix(d2))
        denc = eng.decoder_backward(ctx.c1, fix(d1), denc=denc)
        c = ctx.c1
        return None, None, None, None, denc.view(c["Ts"], c["B"], -1), None, None, None


# ------------------------------------------------------------------ modules
def _init_linear(n, k, gain=1.0):
    w = torch.empty(n, k)
    nn.init.xavier_uniform_(w, gain=gain)
    return w


class ConvolutionalTransformerEncoder(FairseqEncoder):
    """ConvolutionalTransformerEncoder (conv_transformer.py:124-345) on the HIP engine."""

    def __init__(self, args, dictionary, audio_features=40, owner=None):
        super().__init__(dictionary)
        self.args = args
        self.owner = owner
        self.ctc_compress_out = getattr(args, "ctc_compress_out", False)
        if self.ctc_compress_out:
            assert args.criterion == "ctc_multi_loss"          # conv_transformer.py:191
            self.ctc_layer = args.ctc_encoder_layer
        self._last = None

    @property
    def engine(self):
        return self.owner.engine

    @property
    def output_batch_first(self):
        return False                                            # conv_transformer.py:302-304

    def forward(self, src_tokens, src_lengths, cls_input=None, return_all_hiddens=False, want_state=None, **unused):
        m = self.owner
        m._ensure_engine(src_tokens.device)
        if return_all_hiddens and want_state is None and not self.ctc_compress_out:
            want_state = getattr(m, "_ctc_state_layer", None)
        seed = m._next_seed()
        keep = None
        if m.hp.encoder_layerdrop > 0:
            # LayerDrop (conv_transformer.py:238-243): one host draw per layer from torch's CPU generator, in both modes, exactly
            # the reference's `torch.empty(1).uniform_()` sequence -- seeded alike, the same layers drop out
            draws = [float(torch.empty(1).uniform_()) for _ in range(m.hp.enc_layers)]
            if self.training:
                keep = [d > m.hp.encoder_layerdrop for d in draws]
                if m.arena is not None:
                    m.arena.note_layers(["encoder.layers.%d." % l for l in range(m.hp.enc_layers)], keep)
        eo, co, st = _EncoderFn.apply(m.anchor, self, src_tokens, src_lengths, self.training, seed, want_state, keep)
        o = self._last
        B = eo.shape[1]
        pad_mask = None
        if o["klen"] is not None:
            pad_mask = torch.arange(eo.shape[0], device=eo.device)[None, :] >= o["lengths"][:, None]
        states = None
        if return_all_hiddens:
            states = list(o["states"]) if o["states"] is not None else None
            if states is not None and want_state is not None:
                states[want_state] = st
        if self.ctc_compress_out:
            ctc_mask = None
            if o["ctc_klen"] is not None:
                ctc_mask = torch.arange(co.shape[0], device=eo.device)[None, :] >= o["ctc_lengths"][:, None]
            return CTCAwareEncoderOut(eo, pad_mask, None, states, src_tokens, o["lengths"], co, ctc_mask)
        return EncoderOut(eo, pad_mask, None, states, src_tokens, o["lengths"])

    def forward_non_torchscript(self, net_input):
        return self.forward(**{k: v for k, v in net_input.items()
                               if k not in ("prev_output_tokens", "transcript_prev_output_tokens")})   # :306-313

    def reorder_encoder_out(self, encoder_out, new_order):          # conv_transformer.py:315-345
        if encoder_out.encoder_out is not None:
            encoder_out = encoder_out._replace(encoder_out=encoder_out.encoder_out.index_select(1, new_order))
        if encoder_out.encoder_padding_mask is not None:
            encoder_out = encoder_out._replace(encoder_padding_mask=encoder_out.encoder_padding_mask.index_select(0, new_order))
        if encoder_out.src_lengths is not None:
            encoder_out = encoder_out._replace(src_lengths=encoder_out.src_lengths.index_select(0, new_order))
        return encoder_out

    def max_positions(self):
        return getattr(self.args, "max_source_positions", 100000)


class TransformerDecoder(FairseqIncrementalDecoder):
    """TransformerDecoder (fairseq/models/transformer.py:517-866), training / scoring path."""

    def __init__(self, args, dictionary, owner=None, pfx="decoder."):
        super().__init__(dictionary)
        self.args = args
        self.owner = owner
        self.pfx = pfx
        self.padding_idx = dictionary.pad()
        self.max_target_positions = getattr(args, "max_target_positions", 100000)

    @property
    def engine(self):
        return self.owner.engine

    ATTN_PROBS_MAX_KEYS = 2048          # s2t_attn_probs_avg keeps one row of probabilities in LDS (csrc/attention.hip)

    def _alignment_request(self, alignment_layer, alignment_heads, need_attn, src_frames=0):
        """fairseq/models/transformer.py:700-703,756-782: the reference returns the encoder-attention weights of `alignment_layer` (default:
        the last layer), averaged over its first `alignment_heads` heads (default: all), on EVERY forward.  Here they cost an extra
        kernel (the fused attention never materialises P), so they are computed in eval mode -- generation, generate.py
        --print-alignment, ensemble attention averaging -- and in training only when asked for (need_attn=True or an explicit
        alignment_layer); `need_attn=False` switches them off."""
        m = self.owner
        want = need_attn if need_attn is not None else (alignment_layer is not None or not self.training)
        if not want or getattr(m.hp, "dec_layers", 0) == 0:
            return None, None
        if src_frames > self.ATTN_PROBS_MAX_KEYS and not need_attn and alignment_layer is None:
            # nobody asked: validation / generation of a very long utterance (max_source_positions defaults to 100000) goes on
            # without the attention output instead of failing in the kernel that would compute it
            return None, None
        layer = m.hp.dec_layers - 1 if alignment_layer is None else int(alignment_layer)
        return layer, alignment_heads

    def forward(self, prev_output_tokens, encoder_out=None, incremental_state=None, features_only=False, alignment_layer=None,
                alignment_heads=None, need_attn=None, **unused):
        m = self.owner
        m._ensure_engine(prev_output_tokens.device)
        a_layer, a_heads = self._alignment_request(alignment_layer, alignment_heads, need_attn,
                                                   src_frames=int(encoder_out.encoder_out.shape[0]) if encoder_out is not None else 0)
        if incremental_state is not None:
            # transformer.py:690-760 with incremental_state: only the last token is embedded, K/V come from the cache.
            key = "s2t_hip_state.%s" % self.pfx
            st = incremental_state.get(key)
            if st is None:
                st = incremental_state[key] = self.begin_incremental(encoder_out, min(self.max_positions(), 1024))
            assert st["steps"] == prev_output_tokens.shape[1] - 1, "incremental decoding must advance one token at a time"
            st["attn_layer"], st["attn_heads"], st["attn"] = a_layer, a_heads, None
            logits = self.step_incremental(st, prev_output_tokens[:, -1])
            return logits.unsqueeze(1), {"attn": [st["attn"]], "inner_states": None}       # attn [N, 1, Ts] (transformer.py:782)
        eo = encoder_out.encoder_out
        klen = None
        if encoder_out.encoder_padding_mask is not None:
            klen = encoder_out.src_lengths.to(torch.int32)
        keep = None
        if m.hp.decoder_layerdrop > 0:                  # fairseq/modules/layer_drop.py:39-44: one vector of draws per pass
            draws = torch.empty(m.hp.dec_layers).uniform_().tolist()
            if self.training:
                keep = [d > m.hp.decoder_layerdrop for d in draws]
                if m.arena is not None:
                    m.arena.note_layers([self.pfx + "layers.%d." % l for l in range(m.hp.dec_layers)], keep)
        if keep is not None and a_layer is not None and not keep[a_layer]:
            a_layer = None                                        # LayerDrop removed the alignment layer: no attention this pass
        logits_tm, attn = _DecoderFn.apply(m.anchor, self, prev_output_tokens, eo, klen, self.training, m._next_seed(), keep, a_layer, a_heads)
        B, L = prev_output_tokens.shape
        logits = logits_tm.view(L, B, -1).transpose(0, 1)         # view, no copy: (B, L, V)
        return logits, {"attn": [attn if attn.numel() else None], "inner_states": None}     # attn (B, L, Ts), heads averaged

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        """fairseq/models/fairseq_decoder.py:58-79: f32 (log-)probabilities of the decoder's logits, differentiable -- what a criterion
        of the reference other than the three re-registered here consumes (those use the fused loss + gradient kernels and never
        materialise this tensor).  Both directions are kernels (s2t_log_softmax / s2t_softmax_probs, s2t_softmax_bwd)."""
        logits = net_output[0]
        flat = logits.reshape(-1, logits.shape[-1]).contiguous()
        if torch.is_grad_enabled() and flat.requires_grad:
            return K.NormalizedProbs.apply(flat, bool(log_probs)).view(logits.shape)
        return (K.log_softmax(flat) if log_probs else K.softmax_probs(flat)).view(logits.shape)

    # ---- incremental decoding: state = per-hypothesis K/V rows owned by the engine
    def begin_incremental(self, encoder_out, max_steps):
        if self.training:
            raise RuntimeError("incremental decoding requires model.eval()")
        klen = None
        if encoder_out.encoder_padding_mask is not None:
            klen = encoder_out.src_lengths.to(torch.int32)
        with torch.no_grad():
            return self.engine.decoder_begin(encoder_out.encoder_out.contiguous(), klen, max_steps, pfx=self.pfx)

    def step_incremental(self, state, last_tokens):
        """last_tokens int64 [N] -> logits [N, V] (row stride padded)."""
        with torch.no_grad():
            return self.engine.decoder_step(state, last_tokens)

    def reorder_incremental(self, state, new_order):
        self.engine.decoder_reorder(state, new_order)

    def reorder_incremental_state(self, incremental_state, new_order):       # transformer.py:840-852
        st = incremental_state.get("s2t_hip_state.%s" % self.pfx)
        if st is not None:                      # the reference's generator also compacts finished sentences away: encoder side too
            self.engine.decoder_reorder(st, new_order, encoder_side=True)

    def max_positions(self):
        return self.max_target_positions


@register_model("conv_transformer")
class ConvolutionalTransformerModel(FairseqEncoderDecoderModel):
    """ConvolutionalTransformerModel (conv_transformer.py:35-121)."""

    def __init__(self, args, encoder, decoder, hp):
        super().__init__(encoder, decoder)
        self.args = args
        self.hp = hp
        object.__setattr__(encoder, "owner", self)
        object.__setattr__(decoder, "owner", self)
        # parameters live on the CPU (reference initialisation) until materialize() re-homes them in the arena
        self.params = nn.ParameterDict()
        self._names = {}
        for name, shape in hp.param_shapes().items():
            self._register(name, self._init_param(name, shape))
        C = hp.conv_ch
        # BatchNorm buffers: reference state-dict prefix -> (buffer attribute prefix, channels)
        self._bn_refs = {"encoder.bn.%d" % i: ("bn%d" % i, C) for i in range(2)}
        if hp.attn_2d:
            for i in range(2):
                for n, ch in (("bn_q", 4), ("bn_k", 4), ("bn_v", 4), ("bn_out", C)):
                    self._bn_refs["encoder.attn_2d.%d.%s" % (i, n)] = ("a2d%d_%s" % (i, n), ch)
        for attr, ch in self._bn_refs.values():
            self.register_buffer(attr + "_running_mean", torch.zeros(ch))
            self.register_buffer(attr + "_running_var", torch.ones(ch))
            self.register_buffer(attr + "_num_batches_tracked", torch.zeros(1, dtype=torch.int64))
        self.anchor = nn.Parameter(torch.zeros(1), requires_grad=True)      # keeps the autograd bridges alive
        self.anchor._s2t_anchor = True
        self.engine = None
        self.arena = None
        self.compute_dtype = torch.float32
        self._seed_base, self._seed_ctr = 1, 0
        self.extra_param_specs = {}            # e.g. the criterion-owned CTC head, added before materialize()

    # ---- parameter plumbing
    def _register(self, name, tensor):
        key = name.replace(".", "__")
        p = nn.Parameter(tensor)
        # how an optimizer that is handed bare parameters (fairseq/trainer.py:140-146) finds the arena they belong to (fairseq_optim.py)
        p._s2t_name, p._s2t_owner = name, weakref.ref(self)
        self.params[key] = p
        self._names[name] = key

    def _init_param(self, name, shape):
        """Reference initialisation."""
        hp = self.hp
        if name.endswith(("layer_norm.weight", "layernorm_embedding.weight")) or ((".bn." in name or ".bn_" in name) and name.endswith("weight")):
            return torch.ones(shape)
        if name.endswith("bias"):
            return torch.zeros(shape)
        if ".attn_2d." in name:                                  # conv_attention_2d.py:39-44: xavier_uniform on the conv weights
            w = torch.empty(shape)
            nn.init.xavier_uniform_(w)
            return w
        if "convolutions" in name:                               # conv_transformer.py:348-354
            cin = shape[1]
            std = math.sqrt((4 * (1.0 - hp.dropout)) / (3 * cin))
            return torch.randn(shape) * std
        if name.endswith("embed_tokens.weight"):                 # :357-361
            w = torch.randn(shape) * shape[1] ** -0.5
            w[hp.pad] = 0
            return w
        if name.endswith("output_projection.weight"):            # transformer.py:626-631
            return torch.randn(shape) * shape[1] ** -0.5
        if name.endswith("self_attn.qkv.weight"):                # multihead_attention.py:88-106 (gain 1/sqrt2 per matrix)
            D = shape[1]
            return torch.cat([_init_linear(D, D, 1 / math.sqrt(2)) for _ in range(3)], 0)
        if name.endswith("encoder_attn.kv.weight"):
            D = shape[1]
            return torch.cat([_init_linear(D, D, 1 / math.sqrt(2)) for _ in range(2)], 0)
        if name.endswith("encoder_attn.q_proj.weight"):
            return _init_linear(shape[0], shape[1], 1 / math.sqrt(2))
        if name == "encoder.ctc_fc.weight":                      # nn.Linear default init (:190)
            w = torch.empty(shape)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            return w
        return _init_linear(shape[0], shape[1])                  # xavier_uniform (:371-375, transformer_layer.py:395-400)

    def named_arena_params(self):
        return {n: self.params[k] for n, k in self._names.items()}

    def reference_parameter_names(self):
        """`[n for n, p in model.named_parameters()]` of the REFERENCE's module tree for this configuration -- the order in which
        fairseq's trainer hands parameters to its optimizer (fairseq/trainer.py:140-146) and therefore the meaning of the integer
        keys in a reference checkpoint's `last_optimizer_state` (torch.optim state dict).  Registration order of
        ConvolutionalTransformerEncoder.__init__ (conv_transformer.py:124-194), TransformerEncoderLayer / TransformerDecoderLayer
        (fairseq/modules/transformer_layer.py:31-70,167-232: self_attn k/v/q/out, its LayerNorm, [encoder_attn ..], fc1, fc2, final
        LayerNorm), MultiheadAttention (multihead_attention.py:56-75: k_proj, v_proj, q_proj, out_proj), LocalAttention
        (local_attention.py:34-47: one in_proj), ConvAttention2D (conv_attention_2d.py:24-37), TransformerDecoder
        (fairseq/models/transformer.py:540-631: embed_tokens, layers, layer_norm, output_projection).  Pinned to the reference by
        tests/golden/param_order.json."""
        hp = self.hp
        wb = lambda stem: [stem + ".weight", stem + ".bias"]
        names = wb("encoder.convolutions.0") + wb("encoder.convolutions.1")
        if hp.attn_2d:
            for i in range(2):
                a = "encoder.attn_2d.%d." % i
                names += [a + "in_proj_weight", a + "in_proj_bias"] + wb(a + "out_proj")
                for n in ("bn_q", "bn_k", "bn_v", "bn_out"):
                    names += wb(a + n)
        names += wb("encoder.bn.0") + wb("encoder.bn.1") + wb("encoder.fc3")

        def mha(stem):
            out = []
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                out += wb(stem + n)
            return out

        for l in range(hp.enc_layers):
            lp = "encoder.layers.%d." % l
            if hp.distance_penalty:
                names += [lp + "self_attn.in_proj_weight", lp + "self_attn.in_proj_bias"] + wb(lp + "self_attn.out_proj")
            else:
                names += mha(lp + "self_attn.")
            names += wb(lp + "self_attn_layer_norm") + wb(lp + "fc1") + wb(lp + "fc2") + wb(lp + "final_layer_norm")
        names += wb("encoder.layer_norm")
        if hp.layernorm_embedding:                 # registered after layer_norm in the encoder (conv_transformer.py:180-187) ...
            names += wb("encoder.layernorm_embedding")
        if hp.ctc_layer:
            names += wb("encoder.ctc_fc")
        for dec, V in (("decoder.", hp.V_tgt), ("auxiliary_decoder.", hp.V_aux)):
            if V <= 0:
                continue
            names.append(dec + "embed_tokens.weight")
            if hp.layernorm_embedding:             # ... and right behind the embeddings in the decoder (transformer.py:578-581)
                names += wb(dec + "layernorm_embedding")
            for l in range(hp.dec_layers):
                lp = dec + "layers.%d." % l
                names += mha(lp + "self_attn.") + wb(lp + "self_attn_layer_norm") + mha(lp + "encoder_attn.")
                names += wb(lp + "encoder_attn_layer_norm") + wb(lp + "fc1") + wb(lp + "fc2") + wb(lp + "final_layer_norm")
            names += wb(dec + "layer_norm")
            if not hp.share_dec_embed:
                names.append(dec + "output_projection.weight")
        return names

    def materialize(self, device, compute_dtype=torch.float32, extra=None):
        """Move every parameter into one flat arena on `device` (and build the engine)."""
        # extras (criterion-owned heads) go FIRST: backward finalises the arena tail-first, so whatever is never
        # reported as ready must not sit behind the decoder (distributed.BucketedGradReducer)
        named = dict(extra) if extra else {}
        named.update(self.named_arena_params())
        arena = ParamArena({n: tuple(p.shape) for n, p in named.items()}, device, compute_dtype)
        for n, p in named.items():
            arena.p(n).copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = arena.p(n)
            p.grad = arena.g(n)
            p._s2t_owner = weakref.ref(self)             # also on the criterion-owned extras: their loader refreshes the bf16 shadow
        self.anchor.data = self.anchor.data.to(device)
        for b in self.buffers():
            b.data = b.data.to(device)
        arena.refresh_shadow()
        arena.freeze(getattr(self, "_frozen_names", ()))
        # the layer groups LayerDrop can remove from an update (their Adam state keeps its own step count: optim.ArenaAdam.step)
        if self.hp.encoder_layerdrop > 0:
            arena.drop_groups += ["encoder.layers.%d." % l for l in range(self.hp.enc_layers)]
        if self.hp.decoder_layerdrop > 0:
            arena.drop_groups += ["decoder.layers.%d." % l for l in range(self.hp.dec_layers)]
        self.arena, self.compute_dtype = arena, compute_dtype
        self.engine = S2TEngine(self.hp, arena)
        self.engine.bn_buffers = {
            "%s.%s" % (ref, n): getattr(self, "%s_%s" % (attr, n))
            for ref, (attr, _) in self._bn_refs.items() for n in ("running_mean", "running_var", "num_batches_tracked")}
        return arena

    def _ensure_engine(self, device):
        if self.engine is None:
            if torch.device(device).type != "cuda":
                raise RuntimeError("conv_transformer runs on the HIP engine only: move inputs to cuda:N "
                                   "(there is no CPU fallback for the S2T hot path)")
            self.materialize(device, self.compute_dtype)

    def _next_seed(self):
        self._seed_ctr += 1
        return self._seed_base * 1000 + self._seed_ctr

    def set_seed(self, seed):
        """dropout streams are keyed on (seed, call counter): trainer reseeds with seed + num_updates (trainer.py:655-661)"""
        self._seed_base, self._seed_ctr = int(seed) % 1000003, 0

    def set_num_updates(self, num_updates):
        """FairseqTask.train_step calls this before every forward (fairseq_task.py:370-372).  Under the reference's own trainer it
        is where the dropout streams get their per-update seed (its `_set_seed` only seeds torch's generators, which the kernels do
        not read); under this package's Trainer set_seed already ran for this update and the counter must keep running across the
        micro-batches of --update-freq."""
        seed = int(getattr(self.args, "seed", 1)) + int(num_updates)
        if seed % 1000003 != self._seed_base:
            self.set_seed(seed)

    # ---- reference-compatible state dict (split q/k/v, `encoder.bn.N.*` buffers, positional placeholders)
    def _local_attention_keys(self, sd, to_reference):
        """With --distance-penalty the encoder layers hold LocalAttention (modules/local_attention.py:23-47): ONE in_proj_weight
        [3D, D] / in_proj_bias [3D] in q|k|v order = exactly the fused layout of the arena, under another name."""
        if not self.hp.distance_penalty:
            return sd
        out = {}
        for k, v in sd.items():
            if k.startswith("encoder.layers.") and to_reference and k.endswith((".self_attn.qkv.weight", ".self_attn.qkv.bias")):
                k = k.replace("qkv.weight", "in_proj_weight").replace("qkv.bias", "in_proj_bias")
            elif k.startswith("encoder.layers.") and not to_reference and k.endswith((".self_attn.in_proj_weight", ".self_attn.in_proj_bias")):
                k = k.replace("in_proj_weight", "qkv.weight").replace("in_proj_bias", "qkv.bias")
            out[k] = v
        return out

    def state_dict(self, destination=None, prefix="", keep_vars=False):
        sd = {n: p.data.detach().float().cpu().clone() for n, p in self.named_arena_params().items()}
        sd = self._local_attention_keys(sd, True)
        sd = fused_to_reference(sd)
        for ref, (attr, _) in self._bn_refs.items():
            for n in ("running_mean", "running_var", "num_batches_tracked"):
                b = getattr(self, "%s_%s" % (attr, n)).detach().cpu().clone()
                sd["%s.%s" % (ref, n)] = b.view(()) if n == "num_batches_tracked" else b
        if self.hp.share_dec_embed:              # the reference's state dict holds the shared tensor under both names (transformer.py:618-624)
            for dec in ("decoder.", "auxiliary_decoder."):
                if dec + "embed_tokens.weight" in sd:
                    sd[dec + "output_projection.weight"] = sd[dec + "embed_tokens.weight"].clone()
        sd["encoder.embed_positions.embeddings._float_tensor"] = torch.FloatTensor(1)
        sd["decoder.embed_positions._float_tensor"] = torch.FloatTensor(1)
        sd["decoder.version"] = torch.Tensor([3])
        return {prefix + k: v for k, v in sd.items()}

    def load_state_dict(self, state_dict, strict=True, args=None):
        sd = reference_to_fused(self._local_attention_keys({k: v for k, v in state_dict.items()}, False))
        mine = self.named_arena_params()
        missing = [n for n in mine if n not in sd]
        if strict and missing:
            raise RuntimeError("Missing key(s) in state_dict: " + ", ".join(missing))
        with torch.no_grad():
            for n, p in mine.items():
                if n in sd:
                    p.data.copy_(sd[n].to(p.data.dtype))
            for ref, (attr, _) in self._bn_refs.items():
                for n in ("running_mean", "running_var", "num_batches_tracked"):
                    k = "%s.%s" % (ref, n)
                    if k in sd:
                        buf = getattr(self, "%s_%s" % (attr, n))
                        buf.copy_(sd[k].reshape(buf.shape))
        if self.arena is not None:
            self.arena.refresh_shadow()
        if getattr(args, "freeze_pretrained", False):            # conv_transformer.py:114-121: every loaded parameter leaves the optimizer
            self._frozen_names = [n for n in mine if n in sd]
            for n in self._frozen_names:
                mine[n].requires_grad = False
            if self.arena is not None:
                self.arena.freeze(self._frozen_names)
        # what nn.Module.load_state_dict returns and fairseq's trainer inspects (fairseq/trainer.py:207-214): keys of the file that
        # nothing here consumed.  Placeholders of the reference's tree (sinusoidal-table dummies, decoder.version, the shared output
        # projection listed under both names) are known and not "unexpected".
        known = set(mine) | {"%s.%s" % (ref, n) for ref in self._bn_refs for n in ("running_mean", "running_var", "num_batches_tracked")}
        unexpected = [k for k in sd if k not in known and not k.endswith(("_float_tensor", ".version")) and not k.startswith("criterion.")
                      and not (self.hp.share_dec_embed and k.endswith("output_projection.weight"))]
        if strict and unexpected:
            raise RuntimeError("Unexpected key(s) in state_dict: " + ", ".join(unexpected))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def raw_state_dict_upgrade(self, state_dict):                   # conv_transformer.py:105-112
        if self.encoder.ctc_compress_out and "encoder.ctc_fc.weight" not in state_dict["model"]:
            crit = state_dict.get("criterion", {})
            if "ctc_aware_model.fc_out.weight" in crit:
                state_dict["model"]["encoder.ctc_fc.weight"] = crit["ctc_aware_model.fc_out.weight"]
                state_dict["model"]["encoder.ctc_fc.bias"] = crit["ctc_aware_model.fc_out.bias"]
        return state_dict

    # ---- registry surface
    @staticmethod
    def add_args(parser):
        """conv_transformer.py:47-72 + the TransformerModel flags this path reads (transformer.py:95-175)."""
        a = parser.add_argument
        a("--input-feat-per-channel", type=int, metavar="N")
        a("--activation-fn", choices=["relu", "gelu"])
        a("--dropout", type=float, metavar="D"); a("--attention-dropout", type=float, metavar="D")
        a("--activation-dropout", type=float, metavar="D"); a("--relu-dropout", type=float, metavar="D")
        a("--encoder-embed-dim", type=int, metavar="N"); a("--encoder-ffn-embed-dim", type=int, metavar="N")
        a("--encoder-layers", type=int, metavar="N"); a("--encoder-attention-heads", type=int, metavar="N")
        a("--decoder-embed-dim", type=int, metavar="N"); a("--decoder-ffn-embed-dim", type=int, metavar="N")
        a("--decoder-layers", type=int, metavar="N"); a("--decoder-attention-heads", type=int, metavar="N")
        a("--encoder-normalize-before", action="store_true"); a("--decoder-normalize-before", action="store_true")
        a("--share-decoder-input-output-embed", action="store_true")
        a("--encoder-layerdrop", type=float, metavar="D", default=0)      # fairseq/models/transformer.py:160-163
        a("--decoder-layerdrop", type=float, metavar="D", default=0)
        a("--compute-dtype", choices=["fp32", "bf16"], default=None,
          help="build-defined: arithmetic of the HIP engine (bf16 = bf16 storage, f32 accumulation, f32 master weights)")
        a("--no-scale-embedding", action="store_true")
        a("--encoder-convolutions", type=str, metavar="EXPR")
        a("--normalization-constant", type=float, default=1.0)
        # no explicit default: inside fairseq's model-specific group (argument_default=SUPPRESS, options.py:158-164) the attribute is
        # then absent unless the flag is given, so the arch function decides -- ON for the reference's conv_transformer* names
        # (conv_transformer.py:65,455), OFF for the build-defined s2t_transformer* presets; --attn-2d forces it on
        a("--no-attn-2d", action="store_true")
        a("--attn-2d", action="store_true", help="keep the two ConvAttention2D blocks with an s2t_transformer* preset")
        a("--distance-penalty", type=str, default=False, choices=["log", "gauss"])
        a("--ctc-compress-out", action="store_true", default=False)
        a("--ctc-compress-strategy", type=str, default="avg", choices=["avg", "weighted", "softmax"])
        a("--freeze-pretrained", action="store_true")
        a("--init-variance", type=float, default=1.0)                  # conv_transformer.py:66-67 (only read by the gauss penalty)
        # The rest of TransformerModel.add_args (fairseq/models/transformer.py:95-175), which the reference's model inherits: a
        # command line written for the reference must PARSE here; what this path does not implement is refused at build time,
        # by name (`_refuse_unbuilt_options`), instead of being dropped silently.
        a("--encoder-embed-path", type=str, metavar="STR"); a("--decoder-embed-path", type=str, metavar="STR")
        a("--encoder-learned-pos", action="store_true"); a("--decoder-learned-pos", action="store_true")
        a("--decoder-output-dim", type=int, metavar="N")
        a("--share-all-embeddings", action="store_true")
        a("--no-token-positional-embeddings", default=False, action="store_true")
        a("--adaptive-softmax-cutoff", metavar="EXPR"); a("--adaptive-softmax-dropout", type=float, metavar="D")
        a("--layernorm-embedding", action="store_true")
        a("--no-cross-attention", default=False, action="store_true"); a("--cross-self-attention", default=False, action="store_true")
        a("--encoder-layers-to-keep", default=None); a("--decoder-layers-to-keep", default=None)
        a("--quant-noise-pq", type=float, metavar="D", default=0); a("--quant-noise-pq-block-size", type=int, metavar="D", default=8)
        a("--quant-noise-scalar", type=float, metavar="D", default=0)

    @staticmethod
    def _refuse_unbuilt_options(args):
        g = lambda k: getattr(args, k, None)
        bad = [k for k in ("encoder_embed_path", "decoder_embed_path", "encoder_learned_pos", "decoder_learned_pos", "share_all_embeddings",
                           "no_token_positional_embeddings", "adaptive_softmax_cutoff", "no_cross_attention",
                           "cross_self_attention", "encoder_layers_to_keep", "decoder_layers_to_keep", "quant_noise_pq",
                           "quant_noise_scalar") if g(k)]
        if g("decoder_output_dim") not in (None, g("decoder_embed_dim")):
            bad.append("decoder_output_dim != decoder_embed_dim")
        if bad:
            raise NotImplementedError("TransformerModel options outside the S2T hot path (SURVEY.md 2.2): " +
                                      ", ".join("--" + k.replace("_", "-") for k in bad))

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        cls._refuse_unbuilt_options(args)
        if getattr(args, "distributed_world_size", 1) > 1 and getattr(args, "ddp_backend", None) == "c10d" and not getattr(args, "use_bmuf", False):
            # fairseq's DEFAULT wrapper (fairseq/options.py:387, fairseq/models/distributed_fairseq_model.py:31-45) is torch's c10d
            # DistributedDataParallel, whose reducer counts autograd gradients per parameter -- the kernels write the arena's
            # gradients directly, so it would wait forever.  The model is built before fairseq's trainer wraps it
            # (fairseq_cli/train.py:60-80, fairseq/trainer.py:100-112), so the run is moved to the wrapper that reduces `p.grad` in
            # place after backward (legacy_distributed_data_parallel.py:96-170: the reference paper script's --ddp-backend no_c10d).
            import warnings
            warnings.warn("--ddp-backend c10d cannot see gradients written by the HIP kernels; using --ddp-backend no_c10d "
                          "(LegacyDistributedDataParallel: one flat all-reduce after backward, as the reference's S2T recipes run)")
            args.ddp_backend = "no_c10d"
        if not hasattr(args, "max_source_positions"):
            args.max_source_positions = 100000
        if not hasattr(args, "max_target_positions"):
            args.max_target_positions = 100000
        if getattr(args, "distance_penalty", False) is True:
            args.distance_penalty = "log"                                    # conv_transformer.py:160-161
        if getattr(args, "distance_penalty", False) not in (False, None, "log"):
            raise NotImplementedError("--distance-penalty gauss (learnable per-head variance; undefined `num_heads` in the reference, "
                                      "conv_transformer_layer.py:30-38) is not built; `log` is (SURVEY.md 8-f N4)")
        src_dict, tgt_dict = task.source_dictionary, task.target_dictionary
        enc_dict = src_dict if src_dict is not None else tgt_dict           # conv_transformer.py:100-101
        convs = eval(args.encoder_convolutions) if isinstance(args.encoder_convolutions, str) else args.encoder_convolutions
        assert len(convs) == 2 and all(tuple(c[1:]) in ((3,), (3, 3)) for c in convs) and convs[0][0] == convs[1][0], \
            "the subsampler kernels implement 2 x (C, 3, 3) stride-2 convolutions"
        assert args.encoder_embed_dim == args.decoder_embed_dim and args.encoder_attention_heads == args.decoder_attention_heads
        compress = getattr(args, "ctc_compress_out", False)
        hp = HParams(D=args.encoder_embed_dim, heads=args.encoder_attention_heads, ffn=args.encoder_ffn_embed_dim,
                     enc_layers=args.encoder_layers, dec_layers=args.decoder_layers, conv_ch=convs[0][0],
                     feat=args.input_feat_per_channel, ctc_layer=(args.ctc_encoder_layer if compress else 0),
                     ctc_strategy=STRATEGIES[getattr(args, "ctc_compress_strategy", "avg")],
                     act=getattr(args, "activation_fn", "relu"), dropout=args.dropout,
                     attention_dropout=args.attention_dropout, activation_dropout=args.activation_dropout,
                     pad=tgt_dict.pad(), no_scale_embedding=getattr(args, "no_scale_embedding", False),
                     V_src=len(enc_dict), V_tgt=len(tgt_dict), distance_penalty=getattr(args, "distance_penalty", False) or False,
                     attn_2d=bool(getattr(args, "attn_2d", False)),
                     share_dec_embed=bool(getattr(args, "share_decoder_input_output_embed", False)),
                     layernorm_embedding=bool(getattr(args, "layernorm_embedding", False)),
                     encoder_layerdrop=float(getattr(args, "encoder_layerdrop", 0) or 0),
                     decoder_layerdrop=float(getattr(args, "decoder_layerdrop", 0) or 0))
        assert args.decoder_ffn_embed_dim == args.encoder_ffn_embed_dim
        encoder = ConvolutionalTransformerEncoder(args, enc_dict, audio_features=args.input_feat_per_channel)
        decoder = TransformerDecoder(args, tgt_dict)
        return cls(args, encoder, decoder, hp)


@register_model("conv_transformer_dualdecoder")
class ConvolutionalTransformerDualDecoder(ConvolutionalTransformerModel):
    """conv_transformer_dualdecoder.py:13-81 (+ multi_task.py:7-21): one encoder, a translation decoder and an
    auxiliary transcript decoder over the same encoder output."""

    def __init__(self, args, encoder, decoder, hp, auxiliary_decoder):
        super().__init__(args, encoder, decoder, hp)
        self.auxiliary_decoder = auxiliary_decoder
        object.__setattr__(auxiliary_decoder, "owner", self)

    @staticmethod
    def add_args(parser):
        ConvolutionalTransformerModel.add_args(parser)
        parser.add_argument("--auxiliary-decoder-embed-path", type=str, metavar="STR")

    @classmethod
    def build_model(cls, args, task):
        assert task.source_dictionary is not None, "the dual-decoder model needs a task with transcripts (:41)"
        base = ConvolutionalTransformerModel.build_model.__func__(ConvolutionalTransformerModel, args, _TgtOnly(task))
        hp = base.hp
        hp.V_aux = len(task.source_dictionary)
        enc = ConvolutionalTransformerEncoder(args, task.target_dictionary, audio_features=args.input_feat_per_channel)
        dec = TransformerDecoder(args, task.target_dictionary)
        aux = TransformerDecoder(args, task.source_dictionary, pfx="auxiliary_decoder.")
        return cls(args, enc, dec, hp, aux)

    def get_auxiliary_target(self, sample, auxiliary_output):
        return sample["transcript_target"]

    def get_auxiliary_token_lens(self, sample):
        return sample["transcript_target_lengths"]

    def forward(self, src_tokens, src_lengths, prev_output_tokens, transcript_prev_output_tokens, **kwargs):
        self._ensure_engine(src_tokens.device)
        if self.hp.decoder_layerdrop > 0:
            raise NotImplementedError("--decoder-layerdrop with the dual-decoder model (the two decoders would need separate draws)")
        eo = self.encoder(src_tokens, src_lengths=src_lengths)
        klen = eo.src_lengths.to(torch.int32) if eo.encoder_padding_mask is not None else None
        l1, l2 = _DualDecoderFn.apply(self.anchor, self, prev_output_tokens, transcript_prev_output_tokens, eo.encoder_out,
                                      klen, self.training, self._next_seed())
        (B, L1), L2 = prev_output_tokens.shape, transcript_prev_output_tokens.shape[1]
        extra = {"attn": [None], "inner_states": None}
        return (l1.view(L1, B, -1).transpose(0, 1), extra), (l2.view(L2, B, -1).transpose(0, 1), extra)


class _TgtOnly:
    """task view whose source dictionary is hidden: the dual-decoder encoder is built with the TARGET dictionary
    (conv_transformer_dualdecoder.py:59-60)"""

    def __init__(self, task):
        self.target_dictionary = task.target_dictionary
        self.source_dictionary = None


# ------------------------------------------------------------------ architectures
def _common(args):
    args.dropout = getattr(args, "dropout", 0.3)
    args.normalization_constant = getattr(args, "normalization_constant", 0.5)
    args.attention_dropout = getattr(args, "attention_dropout", 0.1)
    # conv_transformer.py:434 (relu_dropout defaults to 0.1) + transformer_layer.py:43-46 (activation_dropout,
    # and when that is 0 the layer falls back to relu_dropout -- also when 0 was asked for explicitly)
    rd = getattr(args, "relu_dropout", None)
    args.relu_dropout = 0.1 if rd is None else rd
    ad = getattr(args, "activation_dropout", None) or 0
    args.activation_dropout = ad if ad != 0 else args.relu_dropout
    args.attn_2d = not getattr(args, "no_attn_2d", False)
    args.no_token_positional_embeddings = getattr(args, "no_token_positional_embeddings", False)
    args.share_decoder_input_output_embed = getattr(args, "share_decoder_input_output_embed", False)
    args.decoder_embed_path = getattr(args, "decoder_embed_path", None)
    args.encoder_learned_pos = getattr(args, "encoder_learned_pos", False)
    args.encoder_normalize_before = True          # conv_transformer.py:450 (store_true flag: cannot be disabled)
    args.decoder_normalize_before = True
    args.distance_penalty = getattr(args, "distance_penalty", False)
    args.decoder_learned_pos = getattr(args, "decoder_learned_pos", False)
    args.no_scale_embedding = getattr(args, "no_scale_embedding", False)
    args.layernorm_embedding = getattr(args, "layernorm_embedding", False)
    args.input_feat_per_channel = getattr(args, "input_feat_per_channel", None) or 80
    args.activation_fn = getattr(args, "activation_fn", None) or "relu"


def _sizes(args, D, Ff, H, EL, DL, conv):
    args.encoder_embed_dim = getattr(args, "encoder_embed_dim", None) or D
    args.encoder_convolutions = getattr(args, "encoder_convolutions", None) or conv
    args.encoder_layers = getattr(args, "encoder_layers", None) or EL
    args.encoder_ffn_embed_dim = getattr(args, "encoder_ffn_embed_dim", None) or Ff
    args.encoder_attention_heads = getattr(args, "encoder_attention_heads", None) or H
    args.decoder_embed_dim = getattr(args, "decoder_embed_dim", None) or D
    args.decoder_layers = getattr(args, "decoder_layers", None) or DL
    args.decoder_ffn_embed_dim = getattr(args, "decoder_ffn_embed_dim", None) or Ff
    args.decoder_attention_heads = getattr(args, "decoder_attention_heads", None) or H
    args.decoder_output_dim = args.decoder_out_embed_dim = args.decoder_embed_dim


@register_model_architecture("conv_transformer", "conv_transformer")
def base_architecture(args):                                       # conv_transformer.py:429-466
    _common(args); _sizes(args, 256, 768, 4, 6, 6, "[(64, 3, 3)] * 2")


@register_model_architecture("conv_transformer", "conv_transformer_big")
def conv_transformer_big(args):                                    # :469-506
    _common(args); _sizes(args, 512, 1024, 8, 6, 6, "[(64, 3, 3)] * 2")


@register_model_architecture("conv_transformer", "conv_transformer_big2")
def conv_transformer_big2(args):                                   # :509-546
    _common(args); _sizes(args, 512, 2048, 8, 6, 6, "[(64, 3, 3)] * 2")


@register_model_architecture("conv_transformer", "conv_transformer_giant")
def conv_transformer_giant(args):                                  # :549-586
    _common(args); _sizes(args, 1024, 4096, 16, 6, 6, "[(128, 3, 3)] * 2")


def _s2t(args, D, Ff, H, EL, DL, p, conv="[(64, 3, 3)] * 2"):
    """Build-defined presets named by BASELINE.json: conv_transformer structure, S2T sizes."""
    args.dropout = getattr(args, "dropout", None) if getattr(args, "dropout", None) is not None else p
    args.no_attn_2d = getattr(args, "no_attn_2d", True) and not getattr(args, "attn_2d", False)   # off unless --attn-2d (BASELINE.md workload)
    _common(args); _sizes(args, D, Ff, H, EL, DL, conv)


@register_model_architecture("conv_transformer", "s2t_transformer")
def s2t_transformer(args):
    _s2t(args, 256, 2048, 4, 12, 6, 0.1)


@register_model_architecture("conv_transformer", "s2t_transformer_s")
def s2t_transformer_s(args):
    _s2t(args, 256, 2048, 4, 12, 6, 0.1)


@register_model_architecture("conv_transformer", "s2t_transformer_xs")
def s2t_transformer_xs(args):
    _s2t(args, 256, 1024, 4, 6, 3, 0.3)


@register_model_architecture("conv_transformer", "s2t_transformer_m")
def s2t_transformer_m(args):
    _s2t(args, 512, 2048, 8, 12, 6, 0.15)


@register_model_architecture("conv_transformer", "s2t_transformer_l")
def s2t_transformer_l(args):
    _s2t(args, 1024, 4096, 16, 12, 6, 0.2, "[(128, 3, 3)] * 2")


@register_model_architecture("conv_transformer_dualdecoder", "conv_transformer_dualdecoder")
def dualdecoder_base(args):                                        # conv_transformer_dualdecoder.py:91-94
    base_architecture(args)
    args.auxiliary_decoder_embed_path = getattr(args, "auxiliary_decoder_embed_path", None)


@register_model_architecture("conv_transformer_dualdecoder", "conv_transformer_dualdecoder_big")
def dualdecoder_big(args):
    conv_transformer_big(args)
    args.auxiliary_decoder_embed_path = getattr(args, "auxiliary_decoder_embed_path", None)


@register_model_architecture("conv_transformer_dualdecoder", "conv_transformer_dualdecoder_big2")
def dualdecoder_big2(args):
    conv_transformer_big2(args)
    args.auxiliary_decoder_embed_path = getattr(args, "auxiliary_decoder_embed_path", None)
