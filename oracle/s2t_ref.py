"""ORACLE (test infrastructure only): fp32 CPU restatement of the S2T hot path.

Functional style: every function takes a flat ``W`` dict whose keys are the
reference's state-dict names and plain tensors.  Only plain
torch CPU ops are used -- the same third-party arithmetic the reference itself
bottoms out in -- plus oracle/int_ref.py for the integer parts.
Each function cites the reference lines it restates.  Pinned against golden
vectors captured from the real reference (tests/golden/, test_oracle_golden.py).
"""
import math
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

from . import int_ref

EncOut = namedtuple("EncOut", "encoder_out encoder_padding_mask src_lengths ctc_out ctc_padding_mask "
                              "ctc_pred new_lengths encoder_states")


def default_cfg(**kw):
    cfg = dict(D=256, heads=4, ffn=768, enc_layers=6, dec_layers=6, ctc_layer=0, act="relu",
               enc_pre_ln=True, dec_pre_ln=True, pad=1, strategy="avg", conv_ch=64, feat=80,
               no_scale_embedding=False, ln_eps=1e-5, bn_eps=1e-5, bn_momentum=0.1, layernorm_embedding=False)
    cfg.update(kw)
    return cfg


def site_act(cfg, site, act):
    """The activation of one site, or -- checker-only -- that site's ReLU with the ACTIVE SET taken from the caller
    (cfg["relu_masks"][site], a bool tensor in this function's layout): y = x * mask, gradient routed by the same mask.
    The fp32 parity tests feed it the decisions the product made on its own pre-activations, the way `pred_override` feeds the
    CTC arg-max: a pre-activation within f32 rounding of zero is then active on both sides or on neither, and what is left
    of the comparison is floating-point arithmetic only."""
    m = (cfg.get("relu_masks") or {}).get(site)
    if m is None or cfg["act"] != "relu":
        return act
    return lambda x: x * m.reshape(x.shape).to(x.dtype)


def act_fn(name):
    # fairseq/utils.py:390-408 ; fairseq/modules/gelu.py:24-25 (gelu in fp32)
    if name == "relu":
        return F.relu
    if name == "gelu":
        return lambda x: F.gelu(x.float()).type_as(x)
    raise ValueError(name)


# ------------------------------------------------------------------ positions
def sinusoid_table(n, dim, padding_idx):
    """fairseq/modules/sinusoidal_positional_embedding.py:36-58: [sin | cos], row pad = 0."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(n, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(n, -1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(n, 1)], dim=1)
    if padding_idx is not None:
        emb[padding_idx, :] = 0
    return emb


def audio_positions(lengths, T):
    """positional_embedding_audio.py:21-27 + utils.make_positions (utils.py:192-202):
    position t+1 for t < len_b, 0 (the padding row) otherwise.  (B,T) int64."""
    t = torch.arange(T).unsqueeze(0)
    valid = t < lengths.unsqueeze(1)
    return (t + 1) * valid


def token_positions(tokens, pad):
    """utils.make_positions: cumsum of non-pad * non-pad + pad."""
    mask = tokens.ne(pad).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad


def length_mask(lengths, T=None):
    """create_mask, conv_transformer.py:293-300: True at padding; None if no padding."""
    T = int(lengths.max()) if T is None else T
    m = torch.arange(T).unsqueeze(0) >= lengths.unsqueeze(1)
    return m if bool(m.any()) else None


# ------------------------------------------------------------------ subsampler (a2-a4)
def batch_norm2d(x, w, b, rm, rv, training, momentum, eps):
    """nn.BatchNorm2d (conv_transformer.py:364-368,212).  Training: biased batch variance
    normalises, running_var gets the unbiased one; statistics include padded frames."""
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        new_rm = (1 - momentum) * rm + momentum * mean.detach()
        new_rv = (1 - momentum) * rv + momentum * var.detach() * (n / max(n - 1, 1))
    else:
        mean, var, new_rm, new_rv = rm, rv, rm, rv
    y = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + eps)
    y = y * w[None, :, None, None] + b[None, :, None, None]
    return y, new_rm, new_rv


ATTN2D_HEADS = 4          # conv_transformer.py:155-157: ConvAttention2D(out_channels, 4, dropout)


def conv_attention_2d(W, p, x, cfg, training, stats):
    """conv_attention_2d.py:46-135 as called by the encoder (query = key = value = x [B,C,T,F], no padding mask,
    dropout = identity): 3x3 conv -> (q, k, v) of `heads` channels each, q scaled by C^-0.5 (head_dim = embed_dim, :21-23),
    BatchNorm + ReLU per projection, then per (batch, head) plane [T,F]:
      time attention       softmax_T'(q k^T) v                       -> [T,F]
      frequency attention  softmax_F'(q^T k) v^T, transposed back    -> [T,F]
    the 2*heads planes are concatenated on the channel axis, 3x3 conv back to C channels, BatchNorm, ReLU."""
    H = ATTN2D_HEADS
    B, C, T, Fq = x.shape
    qkv = F.conv2d(x, W[p + "in_proj_weight"], W[p + "in_proj_bias"], padding=1)
    planes = []
    for i, n in enumerate(("q", "k", "v")):
        z = qkv[:, i * H:(i + 1) * H]
        if n == "q":
            z = z * (C ** -0.5)
        b = p + "bn_%s." % n
        z, rm, rv = batch_norm2d(z.contiguous(), W[b + "weight"], W[b + "bias"], W[b + "running_mean"], W[b + "running_var"],
                                 training, cfg["bn_momentum"], cfg["bn_eps"])
        stats[b + "running_mean"], stats[b + "running_var"] = rm, rv
        planes.append(F.relu(z).reshape(B * H, T, Fq))
    q, k, v = planes
    a_t = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)             # [BH,T,T]
    o_t = torch.bmm(a_t, v)                                                  # [BH,T,F]
    a_f = torch.softmax(torch.bmm(q.transpose(1, 2), k), dim=-1)             # [BH,F,F]  (sums over every frame, padded ones too)
    o_f = torch.bmm(a_f, v.transpose(1, 2)).transpose(1, 2)                  # [BH,T,F]
    cat = torch.cat([o_t.reshape(B, H, T, Fq), o_f.reshape(B, H, T, Fq)], dim=1)
    y = F.conv2d(cat, W[p + "out_proj.weight"], W[p + "out_proj.bias"], padding=1)
    b = p + "bn_out."
    y, rm, rv = batch_norm2d(y, W[b + "weight"], W[b + "bias"], W[b + "running_mean"], W[b + "running_var"],
                             training, cfg["bn_momentum"], cfg["bn_eps"])
    stats[b + "running_mean"], stats[b + "running_var"] = rm, rv
    return F.relu(y)


def subsample(W, cfg, src_tokens, src_lengths, training=False, trace=None):
    """conv_transformer.py:202-232: 2x[conv3x3 s2 p1 -> act -> BN] -> (cfg attn_2d: 2x residual ConvAttention2D,
    :216-222) -> flatten (channel-major) -> fc3 -> act -> + positions.  Dropout is identity (parity mode).
    Returns x (T4,B,D), lengths (B,), dict of updated BN running stats."""
    act = act_fn(cfg["act"])
    x = src_tokens.unsqueeze(1)
    lengths = src_lengths
    stats = {}
    for i in range(2):
        p = "encoder.convolutions.%d." % i
        x = F.conv2d(x, W[p + "weight"], W[p + "bias"], stride=2, padding=1)
        x = site_act(cfg, "encoder.conv%d" % i, act)(x)
        q = "encoder.bn.%d." % i
        x, rm, rv = batch_norm2d(x, W[q + "weight"], W[q + "bias"], W[q + "running_mean"],
                                 W[q + "running_var"], training, cfg["bn_momentum"], cfg["bn_eps"])
        stats[q + "running_mean"], stats[q + "running_var"] = rm, rv
        lengths = torch.ceil(lengths.float() / 2).long()          # :213
        if trace is not None:
            trace["conv%d" % i] = x
    if cfg.get("attn_2d"):
        for i in range(2):
            x = x + conv_attention_2d(W, "encoder.attn_2d.%d." % i, x, cfg, training, stats)               # :216-222
            if trace is not None:
                trace["attn2d%d" % i] = x
    B, C, T4, F4 = x.shape
    x = x.transpose(1, 2).contiguous().view(B, T4, C * F4).transpose(0, 1)   # :225-226
    x = site_act(cfg, "encoder.fc3", act)(F.linear(x, W["encoder.fc3.weight"], W["encoder.fc3.bias"]))    # :227
    if trace is not None:
        trace["fc3"] = x
    table = sinusoid_table(T4 + 1, cfg["D"], 0)
    pos = audio_positions(lengths, T4)
    x = x + table[pos].transpose(0, 1)                                         # :229
    if cfg.get("layernorm_embedding"):                                         # :230-231 (before the dropout of :232)
        x = layer_norm(W, "encoder.layernorm_embedding.", x, cfg["ln_eps"])
    if trace is not None:
        trace["embed"] = x
    return x, lengths, stats


# ------------------------------------------------------------------ attention (a7, a13)
def mha(W, pfx, heads, query, key, key_padding_mask=None, causal=False, dist_penalty=False, probs_out=None):
    """fairseq/modules/multihead_attention.py:108-366 / F.multi_head_attention_forward
    (Appendix B1): q,k,v projections with bias, q * d^-1/2, -inf on padded keys and
    above the diagonal (causal), softmax in fp32, P.V, out-projection.
    query (Tq,B,D), key (Tk,B,D) -> (Tq,B,D)."""
    Tq, B, D = query.shape
    Tk = key.shape[0]
    d = D // heads
    q = F.linear(query, W[pfx + "q_proj.weight"], W[pfx + "q_proj.bias"]) * (d ** -0.5)
    k = F.linear(key, W[pfx + "k_proj.weight"], W[pfx + "k_proj.bias"])
    v = F.linear(key, W[pfx + "v_proj.weight"], W[pfx + "v_proj.bias"])
    q = q.contiguous().view(Tq, B * heads, d).transpose(0, 1)
    k = k.contiguous().view(Tk, B * heads, d).transpose(0, 1)
    v = v.contiguous().view(Tk, B * heads, d).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf")), 1).unsqueeze(0)
    if key_padding_mask is not None:
        s = s.view(B, heads, Tq, Tk).masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        s = s.view(B * heads, Tq, Tk)
    if dist_penalty:
        # LocalAttention + LogPenalty (examples/speech_recognition/modules/local_attention.py:131-133,
        # modules/conv_transformer_layer.py:22-27): scores -= max(0, ln|i - j|), after the padding mask
        dist = (torch.arange(Tq).unsqueeze(1) - torch.arange(Tk).unsqueeze(0)).abs().float()
        s = s - torch.max(torch.zeros_like(dist), torch.log(dist)).unsqueeze(0)
    p = F.softmax(s.float(), dim=-1)
    if probs_out is not None:
        # need_weights / need_head_weights (multihead_attention.py:342-355): the softmax output (before dropout) as
        # (heads, B, Tq, Tk); the decoder averages the first `alignment_heads` of them (transformer.py:772-777)
        probs_out.append(p.view(B, heads, Tq, Tk).transpose(0, 1))
    o = torch.bmm(p, v).transpose(0, 1).contiguous().view(Tq, B, D)
    return F.linear(o, W[pfx + "out_proj.weight"], W[pfx + "out_proj.bias"])


def layer_norm(W, pfx, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), W[pfx + "weight"], W[pfx + "bias"], eps)


def ffn(W, pfx, x, act):
    return F.linear(act(F.linear(x, W[pfx + "fc1.weight"], W[pfx + "fc1.bias"])),
                    W[pfx + "fc2.weight"], W[pfx + "fc2.bias"])


def encoder_layer(W, cfg, pfx, x, pad_mask):
    """fairseq/modules/transformer_layer.py:87-139 (dropout = identity)."""
    act = act_fn(cfg["act"])
    pre = cfg["enc_pre_ln"]
    r = x
    if pre:
        x = layer_norm(W, pfx + "self_attn_layer_norm.", x)
    x = r + mha(W, pfx + "self_attn.", cfg["heads"], x, x, pad_mask, dist_penalty=bool(cfg.get("distance_penalty", False)))
    if not pre:
        x = layer_norm(W, pfx + "self_attn_layer_norm.", x)
    r = x
    if pre:
        x = layer_norm(W, pfx + "final_layer_norm.", x)
    x = r + ffn(W, pfx, x, site_act(cfg, pfx + "ffn", act))
    if not pre:
        x = layer_norm(W, pfx + "final_layer_norm.", x)
    return x


# ------------------------------------------------------------------ CTC compression (a10)
def ctc_compress(W, cfg, x, lengths, pred_override=None):
    """average_same_ctc_features, conv_transformer.py:278-291.
    x (T,B,D) -> x_ctc (T,B,V), compressed (T''max,B,D), new_lengths, pred ids (B,T).
    pred_override (B,T) int, checker-only: take the arg-max from the caller instead of from this function's own f32 logits.
    The reduced-precision (bf16) parity tests use it to split the comparison the way the arithmetic splits: the integer
    path is checked exactly on the product's OWN logits (int_ref), and the float path is checked against this oracle GIVEN
    the same integer path -- a bf16-rounded logit pair within one ulp of a tie may legitimately pick the other index."""
    x_ctc = F.linear(x, W["encoder.ctc_fc.weight"], W["encoder.ctc_fc.bias"])
    with torch.no_grad():
        prob = F.softmax(x_ctc, dim=-1).transpose(0, 1)                  # (B,T,V)  :282
        pred = int_ref.argmax_first_np(prob.numpy()) if pred_override is None else \
            np.ascontiguousarray(np.asarray(pred_override, dtype=np.int64))         # (B,T)    :284
        runs = int_ref.ctc_rle_np(pred, lengths.numpy())                  # :285
        new_lengths = torch.tensor([len(r) for r in runs], dtype=torch.long)
        Wm = torch.from_numpy(int_ref.compress_weights_np(prob.numpy(), runs, cfg["strategy"]))
    out = x.permute(1, 2, 0).bmm(Wm).permute(2, 0, 1)                      # :290-291
    return x_ctc, out, new_lengths, torch.from_numpy(pred.astype(np.int64))


# ------------------------------------------------------------------ encoder (a11)
def encoder_forward(W, cfg, src_tokens, src_lengths, training=False, trace=None, pred_override=None):
    """ConvolutionalTransformerEncoder.forward, conv_transformer.py:195-276."""
    x, lengths, stats = subsample(W, cfg, src_tokens, src_lengths, training, trace)
    mask = length_mask(lengths, x.shape[0])
    x_ctc = ctc_mask = pred = new_lengths = None
    states = []
    keep = cfg.get("enc_keep")            # LayerDrop decisions of this pass (conv_transformer.py:238-243): a dropped layer is skipped,
    for l in range(cfg["enc_layers"]):    # together with the CTC compression and the encoder_states entry that hang off it
        if keep is not None and not keep[l]:
            continue
        x = encoder_layer(W, cfg, "encoder.layers.%d." % l, x, mask)
        if cfg["ctc_layer"] and cfg["ctc_layer"] == l + 1:
            ctc_mask = mask
            x_ctc, x, lengths, pred = ctc_compress(W, cfg, x, lengths, pred_override)
            new_lengths = lengths
            mask = length_mask(lengths, x.shape[0])
        states.append(x)
    if cfg["enc_pre_ln"]:
        x = layer_norm(W, "encoder.layer_norm.", x)
        states[-1] = x
    return EncOut(x, mask, lengths, x_ctc, ctc_mask, pred, new_lengths, states), stats


# ------------------------------------------------------------------ decoder (a12, a13)
def decoder_forward(W, cfg, prev_output_tokens, enc_out, enc_pad_mask, pfx="decoder.", attn_layer=None, attn_heads=None):
    """TransformerDecoder.forward, fairseq/models/transformer.py:636-790 (training path).  attn_layer: also return the
    encoder-attention weights of that layer averaged over its first attn_heads heads, (B, L, Ts) (:756-782) -> (logits, attn)."""
    act = act_fn(cfg["act"])
    D = cfg["D"]
    pad = cfg["pad"]
    B, L = prev_output_tokens.shape
    scale = 1.0 if cfg["no_scale_embedding"] else math.sqrt(D)
    table = sinusoid_table(pad + 1 + L, D, pad)
    x = scale * W[pfx + "embed_tokens.weight"][prev_output_tokens]           # :720
    x = x + table[token_positions(prev_output_tokens, pad)]                   # :728-729
    if cfg.get("layernorm_embedding"):                                         # :731-732
        x = layer_norm(W, pfx + "layernorm_embedding.", x, cfg["ln_eps"])
    x = x.transpose(0, 1)
    self_pad = prev_output_tokens.eq(pad)
    self_pad = self_pad if bool(self_pad.any()) else None                     # :739-741
    pre = cfg["dec_pre_ln"]
    attn = None
    dkeep = cfg.get("dec_keep")           # --decoder-layerdrop (fairseq/modules/layer_drop.py:39-44)
    for l in range(cfg["dec_layers"]):
        if dkeep is not None and not dkeep[l]:
            continue
        p = pfx + "layers.%d." % l
        r = x
        if pre:
            x = layer_norm(W, p + "self_attn_layer_norm.", x)
        x = r + mha(W, p + "self_attn.", cfg["heads"], x, x, self_pad, causal=True)
        if not pre:
            x = layer_norm(W, p + "self_attn_layer_norm.", x)
        r = x
        if pre:
            x = layer_norm(W, p + "encoder_attn_layer_norm.", x)
        probs = [] if l == attn_layer else None
        x = r + mha(W, p + "encoder_attn.", cfg["heads"], x, enc_out, enc_pad_mask, probs_out=probs)
        if probs:
            attn = probs[0][:attn_heads].mean(dim=0) if attn_heads is not None else probs[0].mean(dim=0)
        if not pre:
            x = layer_norm(W, p + "encoder_attn_layer_norm.", x)
        r = x
        if pre:
            x = layer_norm(W, p + "final_layer_norm.", x)
        x = r + ffn(W, p, x, site_act(cfg, p + "ffn", act))
        if not pre:
            x = layer_norm(W, p + "final_layer_norm.", x)
    if pre:
        x = layer_norm(W, pfx + "layer_norm.", x)
    x = x.transpose(0, 1)
    # :784-788; --share-decoder-input-output-embed: the projection's weight is the embedding table itself
    # (fairseq/models/transformer.py:618-624), so its state dict need not carry a second copy
    w_out = W.get(pfx + "output_projection.weight")
    logits = F.linear(x, w_out if w_out is not None else W[pfx + "embed_tokens.weight"])
    return logits if attn_layer is None else (logits, attn)


# ------------------------------------------------------------------ generation (a22)
def _beam_search_sentence(W, cfg, eo, n, beam, max_len, min_len, len_penalty, unk_penalty, temperature, normalize, eos, unk,
                          pfx="decoder.", prev_scores=None, prefix=None, no_repeat_ngram=0):
    """One sentence of SequenceGenerator._generate + BeamSearch.step (fairseq/sequence_generator.py:163-500, fairseq/search.py:50-85),
    without incremental state (the full prefix is re-decoded every step).  prev_scores [beam]: HierarchicalBeamSearch.step
    (twophase_sequence_generator.py:22-49) -- at step 0 EVERY beam slot competes, each starting from its own score.
    W / eo / n may be LISTS (one entry per model of an ensemble, EnsembleModel.forward_decoder :711-770: the models' log-probabilities
    are averaged in probability space, logsumexp - log M); prefix: forced first target tokens of this sentence (1-D, pad = free,
    _prefix_tokens :449-476); no_repeat_ngram: n-gram blocking (_no_repeat_ngram :617-650).
    Returns (tokens, score, positional_scores, origin slot) tuples sorted best first."""
    pad = cfg["pad"]
    Ws, eos_, ns = (W, eo, n) if isinstance(W, (list, tuple)) else ([W], [eo], [n])
    toks = torch.full((beam, 1), eos, dtype=torch.long)
    cum = torch.zeros((beam, 0))
    blacklist = [False] * beam
    origin = list(range(beam))
    fin = []
    for step in range(max_len + 1):
        lps = [torch.log_softmax(decoder_forward(Wi, cfg, toks, ei.expand(ni, beam, ei.shape[2]), None, pfx=pfx)[:, -1, :] / temperature,
                                 dim=-1) for Wi, ei, ni in zip(Ws, eos_, ns)]
        lp = lps[0] if len(lps) == 1 else torch.logsumexp(torch.stack(lps, 0), 0) - math.log(len(lps))
        lp[lp != lp] = -math.inf
        lp[:, pad] = -math.inf
        lp[:, unk] -= unk_penalty
        if step >= max_len:
            keep = lp[:, eos].clone(); lp[:] = -math.inf; lp[:, eos] = keep
        if prefix is not None and step < prefix.numel() and step < max_len:
            t = int(prefix[step])
            if t != pad:                                   # every slot may only continue with the forced token
                keep = lp[:, t].clone(); lp[:] = -math.inf; lp[:, t] = keep
            if t == eos:                                   # :462-475: all slots become copies of the first one
                toks = toks[:1].expand(beam, -1).clone(); cum = cum[:1].expand(beam, -1).clone(); lp = lp[:1].expand(beam, -1).clone()
        elif step < min_len:
            lp[:, eos] = -math.inf
        if no_repeat_ngram > 0 and step + 2 - no_repeat_ngram >= 0:
            for i in range(beam):                          # tokens that would complete an n-gram this hypothesis already contains
                g = toks[i].tolist()
                head = g[len(g) - (no_repeat_ngram - 1):] if no_repeat_ngram > 1 else []
                for j in range(len(g) - no_repeat_ngram + 1):
                    if g[j:j + no_repeat_ngram - 1] == head:
                        lp[i, g[j + no_repeat_ngram - 1]] = -math.inf
        V = lp.shape[1]
        if step == 0:
            cand = lp[0] if prev_scores is None else (lp + prev_scores.view(-1, 1)).reshape(-1)
        else:
            cand = (lp + cum[:, step - 1:step]).reshape(-1)
        k = min(2 * beam, cand.numel() - 1)
        cs, ci = torch.topk(cand, k)
        ctok, cbeam = (ci % V).tolist(), (ci // V).tolist()
        is_eos = [ctok[i] == eos and cs[i].item() != -math.inf for i in range(k)]
        for i in range(min(beam, k)):
            if blacklist[i]:
                is_eos[i] = False
        for i in range(min(beam, k)):
            if is_eos[i] and len(fin) < beam:
                t = torch.cat([toks[cbeam[i], 1:], torch.tensor([eos])])
                ps = torch.cat([cum[cbeam[i], :step], cs[i:i + 1]])
                ps[1:] = ps[1:] - ps[:-1].clone()
                sc = cs[i].item() / (step + 1) ** len_penalty if normalize else cs[i].item()
                fin.append((t, sc, ps, origin[cbeam[i]]))
        if any(is_eos[:beam]) and (len(fin) == beam or step == max_len):
            break
        for i in range(min(beam, k)):
            is_eos[i] = is_eos[i] or blacklist[i]
        order = [i for i in range(k) if not is_eos[i]] + [i for i in range(k) if is_eos[i]]
        pick = order[:beam]
        blacklist = [is_eos[i] for i in pick]
        toks = torch.cat([toks[[cbeam[i] for i in pick]], torch.tensor([[ctok[i]] for i in pick])], dim=1)
        cum = torch.cat([cum[[cbeam[i] for i in pick]][:, :step], torch.stack([cs[i] for i in pick]).view(-1, 1)], dim=1)
        origin = [origin[cbeam[i]] for i in pick]
    idx = sorted(range(len(fin)), key=lambda i: fin[i][1])
    return [fin[i] for i in reversed(idx)]


def beam_search(W, cfg, src_tokens, src_lengths, beam, max_len_a=0.0, max_len_b=200, min_len=1, len_penalty=1.0,
                unk_penalty=0.0, temperature=1.0, normalize=True, eos=2, unk=3, max_positions=1000, prefix_tokens=None,
                no_repeat_ngram_size=0):
    """SequenceGenerator.generate, one sentence at a time.  W: one weight dict or a list of them (ensemble of models of one
    configuration).  Returns per sentence a list of (tokens, score, positional_scores) sorted best first."""
    Ws = list(W) if isinstance(W, (list, tuple)) else [W]
    encs = [encoder_forward(Wi, cfg, src_tokens, src_lengths, training=False)[0] for Wi in Ws]
    B, src_len = src_tokens.shape[0], src_tokens.shape[1]
    max_len = min(int(max_len_a * src_len + max_len_b), max_positions - 1)
    results = []
    for b in range(B):
        ns = [int(e.src_lengths[b]) for e in encs]
        eos_ = [e.encoder_out[:n, b:b + 1] for e, n in zip(encs, ns)]                # this sentence's frames only: no padding mask needed
        hyps = _beam_search_sentence(Ws, cfg, eos_, ns, beam, max_len, min_len, len_penalty, unk_penalty, temperature, normalize, eos, unk,
                                     prefix=None if prefix_tokens is None else prefix_tokens[b], no_repeat_ngram=no_repeat_ngram_size)
        results.append([h[:3] for h in hyps])
    return results


def two_phase_beam_search(W, cfg, src_tokens, src_lengths, beam, max_len_a=0.0, max_len_b=200, min_len=1, len_penalty=1.0,
                          unk_penalty=0.0, temperature=1.0, normalize=True, eos=2, unk=3, max_positions=1000):
    """TwoPhaseSequenceGenerator._generate on the dual-decoder model (twophase_sequence_generator.py:127-170): (1) beam search with
    `auxiliary_decoder.*` -> `beam` transcript hypotheses per sentence, best first (:477-762); (2) target beam search whose slot i
    starts from transcript hypothesis i's (normalised) score (:171-475; the dual-decoder's target decoder does not read the
    transcript, conv_transformer_dualdecoder.py:83-84).  The target length limit uses the longest transcript hypothesis of the
    BATCH as `src_len` (:178,213-218).  Returns per sentence a list of (tokens, score, positional_scores, aux_tokens), best first."""
    enc, _ = encoder_forward(W, cfg, src_tokens, src_lengths, training=False)
    B, src_len = src_tokens.shape[0], src_tokens.shape[1]
    max_len1 = min(int(max_len_a * src_len + max_len_b), max_positions - 1)
    args = (min_len, len_penalty, unk_penalty, temperature, normalize, eos, unk)
    aux, eos_ = [], []
    for b in range(B):
        n = int(enc.src_lengths[b])
        eo = enc.encoder_out[:n, b:b + 1]
        aux.append(_beam_search_sentence(W, cfg, eo, n, beam, max_len1, *args, pfx="auxiliary_decoder."))
        eos_.append((eo, n))
    max_aux_len = max(len(h[0]) for hs in aux for h in hs)
    max_len2 = min(int(max_len_a * max_aux_len + max_len_b), max_positions - 1)
    results = []
    for b in range(B):
        eo, n = eos_[b]
        prev = torch.tensor([h[1] for h in aux[b]], dtype=torch.float32)
        hyps = _beam_search_sentence(W, cfg, eo, n, beam, max_len2, *args, prev_scores=prev)
        results.append([(t, sc, ps, aux[b][o][0]) for (t, sc, ps, o) in hyps])
    return results


# ------------------------------------------------------------------ losses (a14-a16)
def label_smoothed_nll(logits, target, eps, pad):
    """label_smoothed_cross_entropy.py:12-29 with log_softmax in fp32 (fairseq_decoder.py:58-79)."""
    lp = F.log_softmax(logits.float(), dim=-1).view(-1, logits.shape[-1])
    t = target.reshape(-1, 1)
    nll = -lp.gather(1, t)
    smooth = -lp.sum(dim=-1, keepdim=True)
    m = t.eq(pad)
    nll = nll.masked_fill(m, 0.0).sum()
    smooth = smooth.masked_fill(m, 0.0).sum()
    return (1.0 - eps) * nll + (eps / lp.shape[-1]) * smooth, nll


class _CTCLoss(torch.autograd.Function):
    """Summed CTC negative log-likelihood with zero_infinity (CTC_loss.py:143-151), by the
    Graves alpha/beta recursion in log space; gradient w.r.t. the logits is
    softmax - occupancy (what F.log_softmax + F.ctc_loss back-propagate)."""

    @staticmethod
    def forward(ctx, logits, targets, input_lengths, target_lengths, blank):
        T, B, V = logits.shape
        lp = F.log_softmax(logits.double(), dim=-1).numpy()
        grad = np.zeros((T, B, V))
        total = 0.0
        NEG = -np.inf
        for b in range(B):
            Tb, Lb = int(input_lengths[b]), int(target_lengths[b])
            y = [int(v) for v in targets[b, :Lb]]
            ext = [blank]
            for c in y:
                ext += [c, blank]
            S = len(ext)
            la = np.full((Tb, S), NEG)
            la[0, 0] = lp[0, b, ext[0]]
            if S > 1:
                la[0, 1] = lp[0, b, ext[1]]
            for t in range(1, Tb):
                for s in range(S):
                    a = la[t - 1, s]
                    if s >= 1:
                        a = np.logaddexp(a, la[t - 1, s - 1])
                    if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                        a = np.logaddexp(a, la[t - 1, s - 2])
                    la[t, s] = a + lp[t, b, ext[s]]
            ll = la[Tb - 1, S - 1]
            if S > 1:
                ll = np.logaddexp(ll, la[Tb - 1, S - 2])
            if not np.isfinite(ll):
                continue                                            # zero_infinity
            lb = np.full((Tb, S), NEG)
            lb[Tb - 1, S - 1] = lp[Tb - 1, b, ext[S - 1]]
            if S > 1:
                lb[Tb - 1, S - 2] = lp[Tb - 1, b, ext[S - 2]]
            for t in range(Tb - 2, -1, -1):
                for s in range(S):
                    a = lb[t + 1, s]
                    if s + 1 < S:
                        a = np.logaddexp(a, lb[t + 1, s + 1])
                    if s + 2 < S and ext[s + 2] != blank and ext[s + 2] != ext[s]:
                        a = np.logaddexp(a, lb[t + 1, s + 2])
                    lb[t, s] = a + lp[t, b, ext[s]]
            total += -ll
            occ = np.full((Tb, V), NEG)
            for s in range(S):
                occ[:, ext[s]] = np.logaddexp(occ[:, ext[s]], la[:, s] + lb[:, s])
            grad[:Tb, b, :] = np.exp(lp[:Tb, b, :]) - np.exp(occ - lp[:Tb, b, :] - ll)
        ctx.save_for_backward(torch.from_numpy(grad).to(logits.dtype))
        return torch.tensor(total, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


def ctc_loss_sum(logits, targets, input_lengths, target_lengths, blank):
    return _CTCLoss.apply(logits, targets, input_lengths, target_lengths, blank)


def ctc_branch(cfg, ctc_out, ctc_pad_mask, transcript, transcript_lengths, blank, pad):
    """CTCCriterion.forward, CTC_loss.py:101-175 on (T4,B,V) logits.
    Returns loss, errors, total, input_lengths."""
    T, B, _ = ctc_out.shape
    if ctc_pad_mask is None:
        in_len = torch.full((B,), T, dtype=torch.long)
    else:
        in_len = T - ctc_pad_mask.sum(dim=1)                         # mask is (B,T) here
    loss = ctc_loss_sum(ctc_out, transcript, in_len, transcript_lengths, blank)
    lp = F.log_softmax(ctc_out.float(), dim=-1).transpose(0, 1)      # (B,T,V)
    pred = int_ref.argmax_first_np(lp.detach().numpy())
    err, tot = int_ref.ctc_uer_np(pred, in_len.numpy(), transcript.numpy(), transcript_lengths.numpy(), blank)
    return loss, err, tot, in_len


def ctc_multi_loss(W, cfg, sample, eps, ctc_weight, blank, training=False, pred_override=None):
    """CTCMultiLoss.forward, ctc_multi_loss.py:140-168, with the encoder-owned ctc_fc
    (--ctc-compress-out) or the criterion-owned fc_out on encoder_states[k-1].
    Returns loss, sample_size (= the CTC branch's, :168), logging dict, EncOut."""
    ni = sample["net_input"]
    enc, stats = encoder_forward(W, cfg, ni["src_tokens"], ni["src_lengths"], training, pred_override=pred_override)
    logits = decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
    if enc.ctc_out is not None:
        ctc_feat, ctc_mask = enc.ctc_out, enc.ctc_padding_mask
    else:
        k = sample["ctc_encoder_layer"]
        ctc_feat = F.linear(enc.encoder_states[k - 1], W["criterion.ctc_aware_model.fc_out.weight"],
                            W["criterion.ctc_aware_model.fc_out.bias"])
        ctc_mask = enc.encoder_padding_mask
    ctc, err, tot, _ = ctc_branch(cfg, ctc_feat, ctc_mask, sample["transcript_target"],
                                  sample["transcript_target_lengths"], blank, cfg["pad"])
    real, nll = label_smoothed_nll(logits, sample["target"], eps, cfg["pad"])
    loss = ctc_weight * ctc + real
    ctc_ntokens = int(sample["transcript_target_lengths"].sum())
    log = dict(loss=float(loss.detach()), ctc_loss=float(ctc.detach()), nll_loss=float(nll.detach()), ntokens=sample["ntokens"],
               nsentences=sample["target"].shape[0], sample_size=sample["ntokens"],
               ctc_errors=err, ctc_total=tot, nframes=int(ni["src_lengths"].sum()))
    return loss, ctc_ntokens, log, enc, logits, stats


def kd_loss(logits, target, teacher_idx, teacher_logits, lam, tau, pad):
    """CrossEntropyKnowledgeDistillationCriterion.forward, fairseq/criterions/knowledge_distillation.py:44-96 (summed)."""
    V = logits.shape[-1]
    t = target.reshape(-1)
    mask = t.ne(pad)
    loss = logits.new_zeros(())
    if lam > 0:
        lp = F.log_softmax((logits / tau).float(), dim=-1).view(-1, V)
        tp = F.softmax(teacher_logits / tau, dim=-1).view(-1, teacher_logits.shape[-1])
        sel = lp.gather(1, teacher_idx.reshape(-1, teacher_idx.shape[-1]).long())
        loss = loss + lam * (-(sel * tp).sum(-1) * mask.float()).sum()
    if lam < 1:
        lp = F.log_softmax(logits.float(), dim=-1).view(-1, V)
        loss = loss + (1 - lam) * F.nll_loss(lp, t, ignore_index=pad, reduction="sum")
    return loss


def dual_decoder_loss(W, cfg, sample, eps, w_primary=0.5, w_aux=0.5, training=False):
    """ConvolutionalTransformerDualDecoder.forward + CrossEntropyDualDecoder.forward
    (conv_transformer_dualdecoder.py:74-81, cross_entropy_dualdecoder.py:31-59)."""
    ni = sample["net_input"]
    enc, stats = encoder_forward(W, cfg, ni["src_tokens"], ni["src_lengths"], training)
    lg = decoder_forward(W, cfg, ni["prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask)
    la = decoder_forward(W, cfg, ni["transcript_prev_output_tokens"], enc.encoder_out, enc.encoder_padding_mask,
                         pfx="auxiliary_decoder.")
    pl, pn = label_smoothed_nll(lg, sample["target"], eps, cfg["pad"])
    al, an = label_smoothed_nll(la, sample["transcript_target"], eps, cfg["pad"])
    return w_primary * pl + w_aux * al, dict(primary_loss=float(pl.detach()), auxiliary_loss=float(al.detach()),
                                              primary_nll_loss=float(pn.detach()), auxiliary_nll_loss=float(an.detach())), lg, la


# ------------------------------------------------------------------ optimizer (a19)
def clip_grad_norm(grads, max_norm):
    """fairseq/utils.py:253-277."""
    total = torch.norm(torch.stack([torch.norm(g) for g in grads]))
    if max_norm > 0:
        coef = (max_norm / (total + 1e-6)).clamp(max=1)
        grads = [g * coef for g in grads]
    return total, grads


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.98, eps=1e-8, wd=0.0):
    """fairseq/optim/adam.py:147-202 (Appendix B2). Returns new p, m, v."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    denom = v.sqrt() + eps
    step_size = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    if wd != 0:
        p = p - wd * lr * p
    p = p - step_size * m / denom
    return p, m, v


def inverse_sqrt_lr(num_updates, lr, warmup_updates, warmup_init_lr):
    """fairseq/optim/lr_scheduler/inverse_square_root_schedule.py:36-47,66-73."""
    if warmup_updates > 0 and num_updates < warmup_updates:
        return warmup_init_lr + num_updates * (lr - warmup_init_lr) / warmup_updates
    decay = lr * (max(warmup_updates, 1) ** 0.5)
    return decay * (max(num_updates, 1) ** -0.5)


# ------------------------------------------------------------------ deterministic weights
def param_shapes(cfg, V_src, V_tgt, criterion_fc=False, V_aux=0):
    """State-dict names and shapes of conv_transformer (SURVEY.md 8-b [probe])."""
    D, Ff, C = cfg["D"], cfg["ffn"], cfg["conv_ch"]
    f4 = math.ceil(math.ceil(cfg["feat"] / 2) / 2)
    s = {}
    s["encoder.convolutions.0.weight"] = (C, 1, 3, 3); s["encoder.convolutions.0.bias"] = (C,)
    s["encoder.convolutions.1.weight"] = (C, C, 3, 3); s["encoder.convolutions.1.bias"] = (C,)
    for i in range(2):
        for n in ("weight", "bias", "running_mean", "running_var"):
            s["encoder.bn.%d.%s" % (i, n)] = (C,)
    s["encoder.fc3.weight"] = (D, C * f4); s["encoder.fc3.bias"] = (D,)
    if cfg.get("attn_2d"):
        H = ATTN2D_HEADS
        for i in range(2):
            p = "encoder.attn_2d.%d." % i
            s[p + "in_proj_weight"] = (3 * H, C, 3, 3); s[p + "in_proj_bias"] = (3 * H,)
            s[p + "out_proj.weight"] = (C, 2 * H, 3, 3); s[p + "out_proj.bias"] = (C,)
            for bn, n in (("bn_q", H), ("bn_k", H), ("bn_v", H), ("bn_out", C)):
                for f in ("weight", "bias", "running_mean", "running_var"):
                    s[p + bn + "." + f] = (n,)

    def attn(p):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + n + ".weight"] = (D, D); s[p + n + ".bias"] = (D,)

    def ln(p):
        s[p + "weight"] = (D,); s[p + "bias"] = (D,)

    def ff(p):
        s[p + "fc1.weight"] = (Ff, D); s[p + "fc1.bias"] = (Ff,)
        s[p + "fc2.weight"] = (D, Ff); s[p + "fc2.bias"] = (D,)

    if cfg.get("layernorm_embedding"):
        ln("encoder.layernorm_embedding.")
    for l in range(cfg["enc_layers"]):
        p = "encoder.layers.%d." % l
        attn(p + "self_attn."); ln(p + "self_attn_layer_norm."); ff(p); ln(p + "final_layer_norm.")
    if cfg["enc_pre_ln"]:
        ln("encoder.layer_norm.")
    if cfg["ctc_layer"]:
        s["encoder.ctc_fc.weight"] = (V_src, D); s["encoder.ctc_fc.bias"] = (V_src,)
    for dec, V in (("decoder.", V_tgt), ("auxiliary_decoder.", V_aux)):
        if V <= 0:
            continue
        s[dec + "embed_tokens.weight"] = (V, D)
        if cfg.get("layernorm_embedding"):
            ln(dec + "layernorm_embedding.")
        for l in range(cfg["dec_layers"]):
            p = dec + "layers.%d." % l
            attn(p + "self_attn."); ln(p + "self_attn_layer_norm.")
            attn(p + "encoder_attn."); ln(p + "encoder_attn_layer_norm.")
            ff(p); ln(p + "final_layer_norm.")
        if cfg["dec_pre_ln"]:
            ln(dec + "layer_norm.")
        if not cfg.get("share_dec_embed", False):
            s[dec + "output_projection.weight"] = (V, D)
    if criterion_fc:
        s["criterion.ctc_aware_model.fc_out.weight"] = (V_src, D)
        s["criterion.ctc_aware_model.fc_out.bias"] = (V_src,)
    return s


def make_weights(shapes, seed):
    """Deterministic synthetic weights shared by the golden generator and the tests
    (numpy legacy RandomState is stable across versions).  Scales are chosen so that
    activations stay O(1) through the stack."""
    rs = np.random.RandomState(seed)
    W = {}
    for k in sorted(shapes):
        shp = shapes[k]
        if k.endswith("running_var"):
            a = 0.5 + rs.rand(*shp)
        elif k.endswith("running_mean"):
            a = 0.1 * rs.randn(*shp)
        elif "layer_norm" in k or "layernorm_embedding" in k or ".bn." in k or ".bn_" in k:
            a = (1.0 + 0.1 * rs.randn(*shp)) if k.endswith("weight") else 0.1 * rs.randn(*shp)
        elif k.endswith("bias"):
            a = 0.05 * rs.randn(*shp)
        elif "embed_tokens" in k:
            a = rs.randn(*shp) * (shp[1] ** -0.5)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = rs.randn(*shp) * (1.0 / math.sqrt(fan_in))
        W[k] = torch.from_numpy(a.astype(np.float32))
    return W
