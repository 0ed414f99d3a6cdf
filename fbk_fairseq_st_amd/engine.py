"""Forward / backward schedule of the S2T hot path as explicit sequences of HIP kernel launches.

Restates, launch by launch, what the reference executes through torch autograd
(SURVEY.md 3.2 / 8-a): ConvolutionalTransformerEncoder.forward (conv_transformer.py:195-276),
TransformerEncoderLayer / TransformerDecoderLayer.forward (fairseq/modules/transformer_layer.py:87-139,
243-377), TransformerDecoder.extract_features/output_layer (fairseq/models/transformer.py:674-790) and the
corresponding backward passes.  No torch arithmetic is used: every FLOP is a call into libs2t_hip.so;
torch only allocates buffers and does integer index bookkeeping.  Gradients of parameters are
accumulated by the kernels directly into the arena's flat f32 gradient buffer.
"""
import ctypes
import math
import os

import torch

from . import kernels as K
from . import lib as L


A2D_HEADS = 4                # ConvAttention2D(out_channels, 4, dropout) -- conv_transformer.py:155-157


class HParams:
    """Static model description (what conv_transformer.py reads from `args`)."""

    def __init__(self, **kw):
        self.D = 256; self.heads = 4; self.ffn = 768; self.enc_layers = 6; self.dec_layers = 6
        self.conv_ch = 64; self.feat = 80
        self.ctc_layer = 0                       # 0 = no CTC compression inside the encoder
        self.ctc_strategy = 0                    # 0 avg, 1 weighted, 2 softmax
        self.act = "relu"
        self.dropout = 0.0; self.attention_dropout = 0.0; self.activation_dropout = 0.0
        self.sub_dropout = None                  # None -> max(dropout, 0.1) (conv_transformer.py:214)
        self.pad = 1; self.no_scale_embedding = False
        self.V_src = 0; self.V_tgt = 0
        self.distance_penalty = False            # 'log': encoder self-attention scores -= max(0, ln|i-j|) (local_attention.py:131-133)
        self.attn_2d = False                     # two residual ConvAttention2D blocks after the convolutions (conv_transformer.py:155-157,216-222)
        self.V_aux = 0                           # > 0: second decoder `auxiliary_decoder.*` over this vocabulary (dual-decoder model)
        self.layernorm_embedding = False         # LayerNorm on the embedded input of the encoder and of every decoder, before its dropout
        #                                          (conv_transformer.py:184-187,230-231; fairseq/models/transformer.py:578-581,731-732)
        self.share_dec_embed = False             # --share-decoder-input-output-embed: output_projection.weight IS embed_tokens.weight (transformer.py:618-624)
        self.ln_eps = 1e-5; self.bn_eps = 1e-5; self.bn_momentum = 0.1
        self.encoder_layerdrop = 0.0; self.decoder_layerdrop = 0.0    # LayerDrop rates (decided on the host by the model)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    @property
    def F4(self):
        return ((self.feat + 1) // 2 + 1) // 2

    def param_shapes(self):
        """Arena layout in forward order (so that backward finishes contiguous slices last-to-first)."""
        D, Ff, C = self.D, self.ffn, self.conv_ch
        s = {}
        s["encoder.convolutions.0.weight"] = (C, 1, 3, 3); s["encoder.convolutions.0.bias"] = (C,)
        s["encoder.bn.0.weight"] = (C,); s["encoder.bn.0.bias"] = (C,)
        s["encoder.convolutions.1.weight"] = (C, C, 3, 3); s["encoder.convolutions.1.bias"] = (C,)
        s["encoder.bn.1.weight"] = (C,); s["encoder.bn.1.bias"] = (C,)
        if self.attn_2d:
            H = A2D_HEADS
            for i in range(2):
                p = "encoder.attn_2d.%d." % i
                s[p + "in_proj_weight"] = (3 * H, C, 3, 3); s[p + "in_proj_bias"] = (3 * H,)
                for n in ("q", "k", "v"):
                    s[p + "bn_%s.weight" % n] = (H,); s[p + "bn_%s.bias" % n] = (H,)
                s[p + "out_proj.weight"] = (C, 2 * H, 3, 3); s[p + "out_proj.bias"] = (C,)
                s[p + "bn_out.weight"] = (C,); s[p + "bn_out.bias"] = (C,)
        s["encoder.fc3.weight"] = (D, C * self.F4); s["encoder.fc3.bias"] = (D,)
        if self.layernorm_embedding:
            s["encoder.layernorm_embedding.weight"] = (D,); s["encoder.layernorm_embedding.bias"] = (D,)

        def ln(p):
            s[p + ".weight"] = (D,); s[p + ".bias"] = (D,)

        def lin(p, n, k, bias=True):
            s[p + ".weight"] = (n, k)
            if bias:
                s[p + ".bias"] = (n,)

        for l in range(self.enc_layers):
            p = "encoder.layers.%d." % l
            ln(p + "self_attn_layer_norm"); lin(p + "self_attn.qkv", 3 * D, D); lin(p + "self_attn.out_proj", D, D)
            ln(p + "final_layer_norm"); lin(p + "fc1", Ff, D); lin(p + "fc2", D, Ff)
            if self.ctc_layer == l + 1:
                lin("encoder.ctc_fc", self.V_src, D)
        ln("encoder.layer_norm")
        for dec, V in (("decoder.", self.V_tgt), ("auxiliary_decoder.", self.V_aux)):
            if V <= 0:
                continue
            s[dec + "embed_tokens.weight"] = (V, D)
            if self.layernorm_embedding:
                s[dec + "layernorm_embedding.weight"] = (D,); s[dec + "layernorm_embedding.bias"] = (D,)
            for l in range(self.dec_layers):
                p = dec + "layers.%d." % l
                ln(p + "self_attn_layer_norm"); lin(p + "self_attn.qkv", 3 * D, D); lin(p + "self_attn.out_proj", D, D)
                ln(p + "encoder_attn_layer_norm"); lin(p + "encoder_attn.q_proj", D, D); lin(p + "encoder_attn.kv", 2 * D, D)
                lin(p + "encoder_attn.out_proj", D, D)
                ln(p + "final_layer_norm"); lin(p + "fc1", Ff, D); lin(p + "fc2", D, Ff)
            ln(dec + "layer_norm")
            if not self.share_dec_embed:
                lin(dec + "output_projection", V, D, bias=False)
        return s


def sinusoid_table(n, dim, padding_idx, device):
    """Host-computed table, same expression as fairseq/modules/sinusoidal_positional_embedding.py:36-58
    (f32 arange * exp(...), [sin | cos], padding row zero); uploaded once and cached by the caller."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(n, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(n, -1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(n, 1)], dim=1)
    if padding_idx is not None:
        emb[padding_idx, :] = 0
    return emb.to(device)


_TAPS_BY_CLASS = [(0, 0, [(1, 1)]), (0, 1, [(1, 0), (1, 2)]), (1, 0, [(0, 1), (2, 1)]),
                  (1, 1, [(0, 0), (0, 2), (2, 0), (2, 2)])]            # slot order = kTapSlot in subsample.hip
_CLASS_SLOT0 = [0, 1, 3, 5]


def conv2_maps(B, T2, F2, device):
    """Row maps of the implicit-GEMM formulation of Conv2d(3x3, stride 2, pad 1) over channels-last
    pixels (index bookkeeping only; built once per batch shape and cached)."""
    T4, F4 = (T2 + 1) // 2, (F2 + 1) // 2
    t4 = torch.arange(T4).view(T4, 1, 1); b = torch.arange(B).view(1, B, 1); f4 = torch.arange(F4).view(1, 1, F4)
    fwd = []
    for kh in range(3):
        for kw in range(3):
            t2 = 2 * t4 + kh - 1; f2 = 2 * f4 + kw - 1
            ok = (t2 >= 0) & (t2 < T2) & (f2 >= 0) & (f2 < F2)
            pix = (b * T2 + t2) * F2 + f2
            fwd.append(torch.where(ok, pix, torch.full_like(pix, -1)).reshape(-1))
    fwd = torch.stack(fwd).to(torch.int32).contiguous().to(device)          # [9][P2]
    bwd = []
    for (pt, pf, taps) in _TAPS_BY_CLASS:
        t2 = torch.arange(pt, T2, 2).view(1, -1, 1); f2 = torch.arange(pf, F2, 2).view(1, 1, -1)
        bb = torch.arange(B).view(B, 1, 1)
        rows = ((bb * T2 + t2) * F2 + f2).reshape(-1)
        if rows.numel() == 0:
            bwd.append(None)
            continue
        maps = []
        for (kh, kw) in taps:
            t4s = (t2 + 1 - kh) // 2; f4s = (f2 + 1 - kw) // 2
            ok = (t4s >= 0) & (t4s < T4) & (f4s >= 0) & (f4s < F4)
            pos = (t4s * B + bb) * F4 + f4s
            maps.append(torch.where(ok, pos, torch.full_like(pos, -1)).reshape(-1))
        bwd.append((rows.to(torch.int32).contiguous().to(device),
                    torch.stack(maps).to(torch.int32).contiguous().to(device)))
    return dict(fwd=fwd, bwd=bwd, T4=T4, F4=F4)


def attn2d_maps(B, T4, F4, device):
    """Row maps [9][M] of a 3x3 / stride 1 / pad 1 convolution over the pixel rows r = (t*B + b)*F4 + f (index bookkeeping):
    tap j = 3*kh + kw reads pixel (t + kh - 1, f + kw - 1), -1 outside the plane.  The data gradient uses the same maps with the
    taps of the weights mirrored (s2t_a2d_pack_w mode 1)."""
    t = torch.arange(T4).view(T4, 1, 1); b = torch.arange(B).view(1, B, 1); f = torch.arange(F4).view(1, 1, F4)
    maps = []
    for kh in range(3):
        for kw in range(3):
            ts = t + kh - 1; fs = f + kw - 1
            ok = (ts >= 0) & (ts < T4) & (fs >= 0) & (fs < F4) & (b >= 0)
            pix = (ts * B + b) * F4 + fs
            maps.append(torch.where(ok, pix, torch.full_like(pix, -1)).reshape(-1))
    return torch.stack(maps).to(torch.int32).contiguous().to(device)


def _splitk(n_out, k_in, m_tokens):
    """k-slices of a weight-gradient product: enough workgroups to fill 2 per CU, a multiple of 8 when possible (one slice per
    XCD: gemm.hip deals slices to XCDs; the dW kernel folds two slices per workgroup before the f32 atomic pass); measured with
    tools/tn_sweep.py: 8 for the 48/64-tile outputs, 32 for 512x512."""
    tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
    if tiles >= 256:
        return 1
    sk = int(max(1, min(512 // tiles, m_tokens // 256, 32)))
    return sk - sk % 8 if sk >= 8 else sk


class S2TEngine:
    def __init__(self, hp, arena):
        self.hp = hp
        self.A = arena
        self.dtype = arena.compute_dtype
        self.dev = arena.device
        self._tables = {}
        self._maps = {}
        self.bn_buffers = None          # set by the model: dict name -> tensor (running_mean/var, num_batches)
        self.on_grads_ready = None      # callback(prefix): every gradient of parameters named prefix* is final
        # Weight gradients of the Linears are not needed before the optimizer: in bf16 mode linear_bwd QUEUES (dY, X, dW, db) and the
        # queue goes out as ONE grouped launch per decoder / encoder backward (K.wgrad_group: every 256 x 256 tile of every dW owned by
        # one workgroup over all tokens, no split-K atomics; csrc/wgrad_group.hip).  Parameter groups are reported final only after
        # the launch that holds their products (flush_wgrad).
        self.defer_wgrad = True                 # f32 mode too since round 6 (K.wgrad_group -> csrc/wgrad_f32.hip: no split-K atomics there either)
        self._wq, self._wq_ready, self._wq_post = [], [], []     # _wq_post: what must follow the grouped launch (re-ordering a dW)
        # One C call per Transformer layer and direction (csrc/layer.hip: the block schedules below restated in C++, launch for
        # launch): from Python a launch costs 6-12 us of host time, from C ~2.5 us, and with 8 utterances per GPU the host's launch
        # rate is what bounds the update.  bf16 path with deferred weight gradients; the per-kernel schedules below remain the
        # reference the tests hold it to (bit for bit) and the path of f32 mode.
        # (shapes csrc/layer.hip refuses -- D or ffn not a multiple of 8, D not divisible by heads: its `bad()` -- stay on the per-kernel
        # schedule, whose products fall back to the linear_wgrad / 128-wide routes)
        self.composite = arena.compute_dtype == torch.bfloat16 and hp.D % 8 == 0 and hp.ffn % 8 == 0 and hp.D % hp.heads == 0
        self._descs = {}
        self._call = L.LayerCall()
        self._items = (L.WgradProblem * K.WGRAD_GROUP_MAX)()     # weight-gradient products appended by the layer calls
        self._n_items = 0
        self._keep = []                                          # their operands' owners, until the grouped launch
        # None: the encoder's queued weight gradients are computed in ONE launch after its backward pass (best packing of the CUs).
        # k: also after every k-th layer from the top.  The Trainer sets k = enc_layers / 2 when gradients are all-reduced: two
        # launches of six layers pack as well as one of twelve (576 / 2 tiles each over 256 CUs, tail cut), and half of the encoder's
        # gradient bytes then travel over xGMI underneath the lower layers' backward instead of after it.
        self.wgrad_flush_layers = None
        # diagnostic: a dict here collects the ReLU decisions of the next forward passes of the per-kernel (f32) schedule, site ->
        # bool tensor (tests/test_configs_gpu.py hands them to the oracle, so that the fp32 comparison holds no discrete decision)
        self.relu_record = None
        # True: the decoder's queued weight gradients go out at the end of ITS backward (data-parallel runs: ~130 MB of gradients
        # then travel under the encoder's backward).  False: they wait for the encoder's flush and share its launch; measured on one GPU
        # (tools/ab_engine_flag.py flush_decoder_wgrad): 15.63 ms per update against 15.48 with the separate launch -- the one work
        # list packs the decoder's short reductions no better than its own launch does, and 0.4 ms of matrix work moves behind the
        # encoder's backward where nothing overlaps it.  Kept as a switch for that measurement.
        self.flush_decoder_wgrad = True
        self._a2d_prescale = None
        self.a2d_time_mfma = True        # time attention of ConvAttention2D on the MFMA attention kernels (False: the VALU kernels; tests compare the two)
        # LayerNorm backward also writes dropout(dx) for the block that consumes dx (one pass instead of two; identical bits)
        self.fuse_bwd_dropout = True
        if hp.act not in ("relu", "gelu"):
            raise NotImplementedError("activation_fn %s" % hp.act)
        self.act_fwd = K.ACT_RELU if hp.act == "relu" else K.ACT_GELU
        self.act_bwd = K.ACT_RELU_BWD if hp.act == "relu" else K.ACT_GELU_BWD

    # ------------------------------------------------------------------ small helpers
    def table(self, n, pad):
        key = (pad,)
        t = self._tables.get(key)
        if t is None or t.shape[0] < n:
            t = sinusoid_table(max(n, 1024), self.hp.D, pad, self.dev)
            self._tables[key] = t
        return t

    def maps(self, B, T2, F2):
        key = (B, T2, F2)
        m = self._maps.get(key)
        if m is None:
            m = conv2_maps(B, T2, F2, self.dev)
            self._maps[key] = m
        return m

    def a2d_maps(self, B, T4, F4):
        key = ("a2d", B, T4, F4)
        m = self._maps.get(key)
        if m is None:
            m = attn2d_maps(B, T4, F4, self.dev)
            self._maps[key] = m
        return m

    def _ready(self, prefix):
        """every gradient of parameters named prefix* has been computed -- or queued: then the report waits for the flush"""
        if self._wq or self._n_items or self._wq_ready:
            self._wq_ready.append(prefix)
        elif self.on_grads_ready is not None:
            self.on_grads_ready(prefix)

    def flush_wgrad(self):
        """launch the queued weight-gradient products (one grouped kernel) and report the parameter groups that waited for them"""
        if self._n_items:
            # products appended by the layer calls first, then the ones queued from Python, in ONE launch
            n = self._n_items
            for (dy, x, dw, db) in self._wq:
                if n >= K.WGRAD_GROUP_MAX:
                    K.wgrad_group_raw(n, ctypes.addressof(self._items)); n = 0
                p = self._items[n]; n += 1
                p.dY = dy.data_ptr(); p.X = x.data_ptr(); p.dW = dw.data_ptr(); p.db = db.data_ptr() if db is not None else None
                p.n_out = dy.shape[1]; p.n_in = x.shape[1]; p.tokens = dy.shape[0]
                p.ldy = dy.stride(0); p.ldx = x.stride(0); p.ldw = dw.stride(0)
            K.wgrad_group_raw(n, ctypes.addressof(self._items))
            self._n_items, self._wq, self._keep = 0, [], []
        elif self._wq:
            K.wgrad_group(self._wq)
            self._wq = []
        self._keep = []
        post, self._wq_post = self._wq_post, []
        for fn in post:
            fn()
        ready, self._wq_ready = self._wq_ready, []
        if self.on_grads_ready is not None:
            for prefix in ready:
                self.on_grads_ready(prefix)

    def reset_wgrad(self):
        """drop whatever the last backward queued but never launched (an exception between queueing and the flush: the out-of-memory
        case the reference's train_step recovers from by zeroing the gradients and going on, fairseq/trainer.py:392-405) -- stale
        (dY, X, dW) items must not be added to the next update's gradients.  The queue also keeps every queued dY alive until the
        flush (~2.6 GB at the bench shape: DESIGN.md section 4)."""
        self._wq, self._wq_ready, self._wq_post = [], [], []
        self._n_items, self._keep = 0, []

    def out_proj(self, pfx):
        """parameter-name stem of a decoder's output projection: the embedding itself when input and output embeddings are shared
        (its gradient then collects the projection's dW and the embedding scatter in one buffer)"""
        return pfx + ("embed_tokens" if self.hp.share_dec_embed else "output_projection")

    def W(self, n):
        return self.A.w(n)

    def P(self, n):
        return self.A.p(n)

    def G(self, n):
        return self.A.g(n)

    def linear(self, x2d, name, act=K.ACT_NONE, residual=None, aux_out=None, p_drop=0.0, seed=0, bias=True, pad_rows=False):
        w = self.W(name + ".weight")
        out = K.alloc_rows((x2d.shape[0],), w.shape[0], self.dtype, self.dev) if pad_rows else None
        return K.gemm(x2d, w, bias=self.P(name + ".bias") if bias else None, act=act,
                      residual=residual, aux_out=aux_out, p_drop=p_drop, seed=seed, out=out)

    def linear_bwd(self, dy2d, x2d, name, need_dx=True, act=K.ACT_NONE, aux=None, alpha=1.0, dx_out=None,
                   dx_accumulate=False, bias=True):
        """dW += dy^T x ; db += colsum(dy) ; returns dx = epi(dy @ W)."""
        w = self.W(name + ".weight")
        gw = self.G(name + ".weight")
        gb = self.G(name + ".bias") if bias else None
        if self.defer_wgrad and K.wgrad_group_ok(dy2d, x2d):
            self._wq.append((dy2d, x2d, gw, gb))         # the queue keeps dY and X alive until the grouped launch
        else:
            K.linear_wgrad(dy2d, x2d, gw, gb, splitk=_splitk(gw.shape[0], gw.shape[1], dy2d.shape[0]))
        if not need_dx:
            return None
        return K.gemm(dy2d, w, trans_b=True, act=act, aux=aux, alpha=alpha, out=dx_out, accumulate=dx_accumulate)

    # ------------------------------------------------------------------ subsampler
    def subsample_fwd(self, src_tokens, len4_32, training, seed):
        hp, C = self.hp, self.hp.conv_ch
        gelu = hp.act == "gelu"          # --activation-fn also drives the subsampler (conv_transformer.py:140-142,212,227)
        act = K.ACT_GELU if gelu else K.ACT_RELU
        B, T, F = src_tokens.shape
        x = src_tokens if src_tokens.dtype == torch.float32 else src_tokens.float()
        x = x.contiguous()
        T2, F2 = (T + 1) // 2, (F + 1) // 2
        mp = self.maps(B, T2, F2)
        T4, F4 = mp["T4"], mp["F4"]
        bufs = self.bn_buffers
        p_sub = (max(hp.dropout, 0.1) if hp.sub_dropout is None else hp.sub_dropout) if training else 0.0
        c = dict(x=x, B=B, T=T, F=F, T2=T2, F2=F2, T4=T4, F4=F4, training=training, p_sub=p_sub, seed=seed)
        # conv1 + BN1
        y1, sums1, pre1 = K.conv1_fwd(x, self.P("encoder.convolutions.0.weight"), self.P("encoder.convolutions.0.bias"), C, self.dtype, act)
        cnt1 = B * T2 * F2
        mean1, rstd1, sc1, sh1 = K.bn_finalize(sums1, self.P("encoder.bn.0.weight"), self.P("encoder.bn.0.bias"),
                                               bufs["encoder.bn.0.running_mean"], bufs["encoder.bn.0.running_var"],
                                               bufs["encoder.bn.0.num_batches_tracked"], cnt1, training, hp.bn_momentum, hp.bn_eps)
        y1n = K.bn_apply(y1, sc1, sh1, p_sub, seed + 1)
        # conv2 as implicit GEMM + BN2
        w2p = K.permute_conv_w(self.P("encoder.convolutions.1.weight"), torch.empty((C, 9 * C), dtype=self.dtype, device=self.dev), C, C, 0)
        P2 = T4 * B * F4
        direct = K.conv2_fwd(y1n, w2p, self.P("encoder.convolutions.1.bias"), B, T2, F2, C, act) if y1n.dtype == torch.bfloat16 else None
        if direct is not None:                   # bf16, 64 channels: input rows staged once in LDS (csrc/conv2.hip)
            z2, pre2 = direct
        else:                                    # other shapes / f32: implicit GEMM over per-tap row maps
            pre2 = torch.empty((P2, C), dtype=self.dtype, device=self.dev) if gelu else None     # GELU's backward needs the pre-activation
            z2 = K.gemm(y1n.view(-1, C), w2p, M=P2, K=9 * C, map_a=mp["fwd"], period_a=C,
                        bias=self.P("encoder.convolutions.1.bias"), act=act, aux_out=pre2)
        sums2 = K.chan_sums(z2, C)
        mean2, rstd2, sc2, sh2 = K.bn_finalize(sums2, self.P("encoder.bn.1.weight"), self.P("encoder.bn.1.bias"),
                                               bufs["encoder.bn.1.running_mean"], bufs["encoder.bn.1.running_var"],
                                               bufs["encoder.bn.1.num_batches_tracked"], P2, training, hp.bn_momentum, hp.bn_eps)
        z2n = K.bn_apply(z2, sc2, sh2, p_sub, seed + 2)
        a2d = []
        if hp.attn_2d:                           # x = x + ConvAttention2D(x), twice (conv_transformer.py:216-222)
            for i in range(2):
                z2n, ci = self.attn2d_block_fwd(i, z2n, B, T4, F4, training, seed + 10 + 3 * i)
                a2d.append(ci)
        c["a2d"] = a2d
        # fc3 on channels-last rows: weight columns re-ordered k = c*F4+f -> k' = f*C+c
        w3p = K.permute_cf(self.P("encoder.fc3.weight"), torch.empty((hp.D, F4 * C), dtype=self.dtype, device=self.dev), hp.D, C, F4, 0)
        pre3 = torch.empty((T4 * B, hp.D), dtype=self.dtype, device=self.dev) if gelu else None
        h3 = K.gemm(z2n.view(T4 * B, F4 * C), w3p, bias=self.P("encoder.fc3.bias"), act=act, aux_out=pre3)
        p = hp.dropout if training else 0.0
        lne = None
        if hp.layernorm_embedding:               # positions, LayerNorm, dropout (conv_transformer.py:228-232): the mask moves behind the LayerNorm
            xp = K.add_pos(h3.view(T4, B, hp.D), self.table(T4 + 1, 0), len4_32, out=torch.empty((T4, B, hp.D), dtype=self.dtype, device=self.dev))
            xn, m_, r_ = K.layernorm_fwd(xp.view(T4 * B, hp.D), self.P("encoder.layernorm_embedding.weight"),
                                         self.P("encoder.layernorm_embedding.bias"), hp.ln_eps)
            lne = dict(x=xp.view(T4 * B, hp.D), mean=m_, rstd=r_)
            xe = (K.dropout(xn, p, seed + 3) if p > 0 else xn).view(T4, B, hp.D)
        else:
            xe = K.add_pos(h3.view(T4, B, hp.D), self.table(T4 + 1, 0), len4_32, out=torch.empty((T4, B, hp.D), dtype=self.dtype, device=self.dev),
                           p_drop=p, seed=seed + 3)
        if self.relu_record is not None and not gelu:           # reference layouts: (B,C,T2,F2), (B,C,T4,F4), (T4,B,D)
            self.relu_record["encoder.conv0"] = (y1.view(B, T2, F2, C) > 0).permute(0, 3, 1, 2)
            self.relu_record["encoder.conv1"] = (z2.view(T4, B, F4, C) > 0).permute(1, 3, 0, 2)
            self.relu_record["encoder.fc3"] = (h3 > 0).view(T4, B, hp.D)
        c.update(y1=y1, y1n=y1n, z2=z2, z2n=z2n, h3=h3, w2p=w2p, w3p=w3p, mean1=mean1, rstd1=rstd1, mean2=mean2,
                 rstd2=rstd2, cnt1=cnt1, P2=P2, p=p, pre1=pre1, pre2=pre2, pre3=pre3, lne=lne)
        return xe, c

    def subsample_bwd(self, c, dx):
        """dx: gradient w.r.t. the subsampler output [T4*B, D]."""
        hp, C = self.hp, self.hp.conv_ch
        B, T4, F4 = c["B"], c["T4"], c["F4"]
        mp = self.maps(B, c["T2"], c["F2"])
        dx = dx.contiguous()
        p3 = c["p"]
        if c.get("lne") is not None:             # back through dropout and the embedding LayerNorm first; the activation's gradient then has no mask
            e = c["lne"]
            if p3 > 0:
                dx = K.dropout(dx, p3, c["seed"] + 3)
            dx = K.layernorm_bwd(dx.view(-1, hp.D), e["x"], e["mean"], e["rstd"], self.P("encoder.layernorm_embedding.weight"),
                                 self.G("encoder.layernorm_embedding.weight"), self.G("encoder.layernorm_embedding.bias"))
            p3 = 0.0
        dh3 = K.act_bwd(dx, c["h3"], 1, p3, c["seed"] + 3) if c["pre3"] is None else K.act_bwd(dx, c["pre3"], 2, p3, c["seed"] + 3)
        # fc3: weight gradient in the re-ordered layout, then scattered back (+=) to the master layout
        z2n2d = c["z2n"].view(T4 * B, F4 * C)
        if self.defer_wgrad and K.wgrad_group_ok(dh3, z2n2d):
            # with the Transformer blocks' products in the grouped launch (10 more tiles of the same reduction length); bias included
            gw3p = torch.zeros((hp.D, F4 * C), dtype=torch.float32, device=self.dev)
            self._wq.append((dh3, z2n2d, gw3p, self.G("encoder.fc3.bias")))
            self._wq_post.append(lambda: K.permute_cf(gw3p, self.G("encoder.fc3.weight"), hp.D, C, F4, 1))
        else:
            gw3p = K.gemm(dh3, z2n2d, trans_a=True, trans_b=True, out_dtype=torch.float32, accumulate=True,
                          splitk=_splitk(hp.D, F4 * C, dh3.shape[0]))
            K.permute_cf(gw3p, self.G("encoder.fc3.weight"), hp.D, C, F4, 1)
            K.colsum(dh3, self.G("encoder.fc3.bias"))
        if c["a2d"]:
            dz2n = K.gemm(dh3, c["w3p"], trans_b=True).view(-1, C)
            for ci in reversed(c["a2d"]):
                dz2n = self.attn2d_block_bwd(ci, dz2n)
            if c["p_sub"] > 0:
                K.dropout(dz2n, c["p_sub"], c["seed"] + 2, out=dz2n)
        else:                                   # the dropout mask of z2n rides on the epilogue of the data-gradient product
            dz2n = K.gemm(dh3, c["w3p"], trans_b=True, p_drop=c["p_sub"], seed=c["seed"] + 2).view(-1, C)
        # BN2 backward (+ ReLU mask) -> gradient w.r.t. conv2 + bias
        s2 = K.chan_sums(c["z2"], C, dyn=dz2n, mean=c["mean2"], rstd=c["rstd2"])
        dpre2 = K.bn_bwd_apply(dz2n, c["z2"], c["mean2"], c["rstd2"], self.P("encoder.bn.1.weight"), s2,
                               self.G("encoder.bn.1.weight"), self.G("encoder.bn.1.bias"), c["P2"], c["training"], pre=c["pre2"])
        K.colsum(dpre2, self.G("encoder.convolutions.1.bias"))
        # conv2 weight gradient: 9 gathered TN GEMMs (one per tap) into [Co][tap*Ci+ci], then back to [Co][Ci][3][3]
        y1n2d = c["y1n"].view(-1, C)
        gw2p = torch.zeros((C, 9 * C), dtype=torch.float32, device=self.dev)
        if not K.conv2_wgrad(dpre2, y1n2d, gw2p, B, c["T2"], c["F2"], C):        # one pass, all taps (bf16, 64 channels)
            sk = int(max(1, min(512, c["P2"] // 1024)))       # K = all output pixels: ~16 k-tiles per workgroup
            for tap in range(9):
                K.gemm(dpre2, y1n2d, trans_a=True, trans_b=True, K=c["P2"], out=gw2p[:, tap * C:(tap + 1) * C],
                       accumulate=True, splitk=sk, map_b=mp["fwd"][tap])
        K.permute_conv_w(gw2p, self.G("encoder.convolutions.1.weight"), C, C, 2)
        # conv2 data gradient: one gathered GEMM per input-pixel parity class, scattered to the class's pixels
        # every input pixel belongs to exactly one parity class: the four products write all of dy1n, each with the dropout mask of
        # y1n in its epilogue (the mask index follows the scattered output row).  (One product per pixel-row parity over 2x2 input
        # blocks, N = 2C, was measured too: 282 us against 261 us for the four class products -- 44 % of its MFMA work is zeros.)
        dy1n = torch.empty_like(c["y1n"]).view(-1, C)
        w2q = K.permute_conv_w(self.P("encoder.convolutions.1.weight"), torch.empty((C, 9 * C), dtype=self.dtype, device=self.dev), C, C, 1)
        # bf16, 64 channels: one direct kernel for all four classes (csrc/conv2.hip); otherwise the gathered products
        if not (dpre2.dtype == torch.bfloat16 and K.conv2_dgrad(dpre2, w2q, dy1n, B, c["T2"], c["F2"], C, c["p_sub"], c["seed"] + 1)):
            for ci, (pt, pf, taps) in enumerate(_TAPS_BY_CLASS):
                if mp["bwd"][ci] is None:
                    continue
                rows, maps = mp["bwd"][ci]
                s0, nt = _CLASS_SLOT0[ci], len(taps)
                K.gemm(dpre2, w2q[:, s0 * C:(s0 + nt) * C], M=rows.numel(), K=nt * C, map_a=maps, period_a=C, map_c=rows, out=dy1n,
                       p_drop=c["p_sub"], seed=c["seed"] + 1)
        s1 = K.chan_sums(c["y1"], C, dyn=dy1n, mean=c["mean1"], rstd=c["rstd1"])
        # BatchNorm backward, activation derivative and the conv1 weight / bias sums in one pass (dpre1 is never materialised)
        K.conv1_bwd_bn(c["x"], dy1n, c["y1"].view(-1, C), c["mean1"], c["rstd1"], self.P("encoder.bn.0.weight"), s1,
                       self.G("encoder.convolutions.0.weight").view(C, 9), self.G("encoder.convolutions.0.bias"),
                       self.G("encoder.bn.0.weight"), self.G("encoder.bn.0.bias"), c["cnt1"], c["training"],
                       pre=None if c["pre1"] is None else c["pre1"].view(-1, C))

    # ------------------------------------------------------------------ ConvAttention2D
    def _a2d_bn(self, p, names, sums, count, training):
        """finalise the BatchNorms `names` of block p from grouped sums; returns (mean, rstd, scale, shift) over the concatenated channels"""
        hp, bufs = self.hp, self.bn_buffers
        parts = []
        off = 0
        for n in names:
            g, b = self.P(p + n + ".weight"), self.P(p + n + ".bias")
            Cg = g.numel()
            parts.append(K.bn_finalize(sums[off:off + 2 * Cg] if sums is not None else None, g, b, bufs[p + n + ".running_mean"],
                                       bufs[p + n + ".running_var"], bufs[p + n + ".num_batches_tracked"], count, training,
                                       hp.bn_momentum, hp.bn_eps))
            off += 2 * Cg
        if len(parts) == 1:
            return parts[0]
        return tuple(torch.cat([pt[k] for pt in parts]) for k in range(4))       # copies of 4-element vectors

    def attn2d_block_fwd(self, i, x, B, T4, F4, training, seed):
        """x [M, C] channels-last pixel rows (t, b, f) -> x + relu(bn_out(conv(cat(time attention, frequency attention))))
        (conv_attention_2d.py:46-135 with query = key = value = x, no padding mask)."""
        hp, C, H = self.hp, self.hp.conv_ch, A2D_HEADS
        p = "encoder.attn_2d.%d." % i
        M = x.shape[0]
        mp = self.a2d_maps(B, T4, F4)
        pd = hp.dropout if training else 0.0                        # ConvAttention2D(.., dropout=self.dropout) on both attention maps
        if self._a2d_prescale is None:                              # q *= head_dim^-0.5 with head_dim = embed_dim = C (:21-23,82), before bn_q
            self._a2d_prescale = torch.tensor([C ** -0.5] * H + [1.0] * (16 - H), dtype=torch.float32, device=self.dev)
        ps = self._a2d_prescale
        w_in = K.a2d_pack_w(self.P(p + "in_proj_weight"), 16, C, self.dtype, 0)             # [16, 9C], rows 12..15 zero
        b16 = torch.zeros(16, dtype=torch.float32, device=self.dev)
        b16[:3 * H].copy_(self.P(p + "in_proj_bias"))
        z = K.gemm(x, w_in, M=M, K=9 * C, map_a=mp, period_a=C, bias=b16)                    # [M, 16]
        sums = K.a2d_chan_stats(z, 3 * H, H, prescale=ps) if training else None
        bn_qkv = self._a2d_bn(p, ("bn_q", "bn_k", "bn_v"), sums, M, training)
        qkv = K.a2d_bn_act(z, 3 * H, bn_qkv[2], bn_qkv[3], prescale=ps)
        cat = torch.empty((M, 2 * H), dtype=self.dtype, device=self.dev)
        pl = o_pl = None
        if self.a2d_time_mfma:
            # time attention on the MFMA attention kernels: planes [3][T][B][4 heads x 32 columns] (F4 <= 32 real), head_dim 32, scale 1
            pl = K.a2d_planes(qkv, torch.empty((3, T4, B, 32 * H), dtype=self.dtype, device=self.dev), 0, B, T4, F4, True)
            o_pl, lse = K.attn_fwd(pl[0], pl[1], pl[2], H, scale=1.0, p_drop=pd, seed=seed + 1)
            K.a2d_planes(cat, o_pl.view(1, T4, B, 32 * H), 0, B, T4, F4, False)
        else:
            lse = K.a2d_time_fwd(qkv, cat, B, T4, F4, pd, seed + 1)
        A = K.a2d_freq_fwd(qkv, cat, B, T4, F4, pd, seed + 2)
        w_out = K.a2d_pack_w(self.P(p + "out_proj.weight"), C, 2 * H, self.dtype, 0)          # [C, 72]
        y = K.gemm(cat, w_out, M=M, K=18 * H, map_a=mp, period_a=2 * H, bias=self.P(p + "out_proj.bias"))
        sums_o = K.a2d_chan_stats(y, C, C) if training else None
        bn_o = self._a2d_bn(p, ("bn_out",), sums_o, M, training)
        out = K.a2d_bn_act(y, C, bn_o[2], bn_o[3], res=x)
        ctx = dict(p=p, x=x, z=z, qkv=qkv, cat=cat, lse=lse, A=A, y=y, bn_qkv=bn_qkv, bn_o=bn_o, B=B, T4=T4, F4=F4, pd=pd, seed=seed,
                   training=training, pl=pl, o_pl=o_pl)
        return out, ctx

    def attn2d_block_bwd(self, c, dout):
        """dout [M, C]: gradient w.r.t. the block output x + f(x); returns the gradient w.r.t. x."""
        hp, C, H = self.hp, self.hp.conv_ch, A2D_HEADS
        p, B, T4, F4 = c["p"], c["B"], c["T4"], c["F4"]
        M = dout.shape[0]
        mp = self.a2d_maps(B, T4, F4)
        ps = self._a2d_prescale
        sk = int(max(1, min(512, M // 1024)))
        dout = dout.contiguous()
        # bn_out + ReLU
        s_o = K.a2d_chan_stats(c["y"], C, C, dy=dout, bn=c["bn_o"])
        K.a2d_param_grads(s_o, self.G(p + "bn_out.weight"), self.G(p + "bn_out.bias"))
        dy = K.a2d_bn_bwd(dout, c["y"], C, C, c["bn_o"], s_o, M, c["training"])
        # out_proj convolution
        K.colsum(dy, self.G(p + "out_proj.bias"))
        if not K.a2d_conv_wgrad(dy, c["cat"], self.G(p + "out_proj.weight"), B, T4, F4):       # one pass, all taps
            gwo = torch.zeros((C, 18 * H), dtype=torch.float32, device=self.dev)
            for tap in range(9):
                K.gemm(dy, c["cat"], trans_a=True, trans_b=True, K=M, out=gwo[:, tap * 2 * H:(tap + 1) * 2 * H], accumulate=True,
                       splitk=sk, map_b=mp[tap])
            K.a2d_unpack_wgrad(gwo, self.G(p + "out_proj.weight"), 2 * H)
        w_out_d = K.a2d_pack_w(self.P(p + "out_proj.weight"), 2 * H, C, self.dtype, 1)        # [8, 9C]
        dcat = K.gemm(dy, w_out_d, M=M, K=9 * C, map_a=mp, period_a=C)
        # the two attentions
        dqkv = torch.zeros_like(c["qkv"])
        if c["pl"] is not None:
            pl = c["pl"]
            do_pl = K.a2d_planes(dcat, torch.empty((1, T4, B, 32 * H), dtype=self.dtype, device=self.dev), 0, B, T4, F4, True)
            dpl = torch.empty_like(pl)
            K.attn_bwd(pl[0], pl[1], pl[2], c["o_pl"], do_pl[0], c["lse"], H, dpl[0], dpl[1], dpl[2], scale=1.0, p_drop=c["pd"],
                       seed=c["seed"] + 1)
            K.a2d_planes(dqkv, dpl, 0, B, T4, F4, False)
        else:
            K.a2d_time_bwd(c["qkv"], c["cat"], dcat, c["lse"], dqkv, B, T4, F4, c["pd"], c["seed"] + 1)
        K.a2d_freq_bwd(c["qkv"], dcat, c["A"], dqkv, B, T4, F4, c["pd"], c["seed"] + 2)
        # bn_q / bn_k / bn_v + ReLU
        s_q = K.a2d_chan_stats(c["z"], 3 * H, H, prescale=ps, dy=dqkv, bn=c["bn_qkv"])
        for g, n in enumerate(("bn_q", "bn_k", "bn_v")):
            K.a2d_param_grads(s_q[2 * H * g:2 * H * (g + 1)], self.G(p + n + ".weight"), self.G(p + n + ".bias"))
        dz = K.a2d_bn_bwd(dqkv, c["z"], 3 * H, H, c["bn_qkv"], s_q, M, c["training"], prescale=ps)
        # in_proj convolution (+ the residual branch of x + f(x))
        K.colsum(dz[:, :3 * H], self.G(p + "in_proj_bias"))
        if not K.a2d_conv_wgrad(dz, c["x"], self.G(p + "in_proj_weight"), B, T4, F4):
            gwi = torch.zeros((16, 9 * C), dtype=torch.float32, device=self.dev)
            for tap in range(9):
                K.gemm(dz, c["x"], trans_a=True, trans_b=True, K=M, out=gwi[:, tap * C:(tap + 1) * C], accumulate=True, splitk=sk,
                       map_b=mp[tap])
            K.a2d_unpack_wgrad(gwi, self.G(p + "in_proj_weight"), C)
        w_in_d = K.a2d_pack_w(self.P(p + "in_proj_weight"), C, 16, self.dtype, 1)            # [C, 144]
        return K.gemm(dz, w_in_d, M=M, K=144, map_a=mp, period_a=16, residual=dout)

    # ------------------------------------------------------------------ transformer blocks
    def self_attn_block_fwd(self, pfx, x, klen32, causal, training, seed, dist_penalty=False):
        """x [T,B,D] -> x + dropout(out_proj(attn(LN(x))))   (pre-LN; transformer_layer.py:103-124)"""
        hp = self.hp
        T, B, D = x.shape
        x2 = x.view(T * B, D)
        h, mean, rstd = K.layernorm_fwd(x2, self.P(pfx + "self_attn_layer_norm.weight"), self.P(pfx + "self_attn_layer_norm.bias"), hp.ln_eps)
        qkv = self.linear(h, pfx + "self_attn.qkv").view(T, B, 3 * D)
        pa = hp.attention_dropout if training else 0.0
        ctx, lse = K.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], hp.heads, klen=klen32, causal=causal,
                              p_drop=pa, seed=seed + 1, dist_penalty=dist_penalty)
        p = hp.dropout if training else 0.0
        y = self.linear(ctx.view(T * B, D), pfx + "self_attn.out_proj", residual=x2, p_drop=p, seed=seed + 2)
        c = dict(x=x2, h=h, mean=mean, rstd=rstd, qkv=qkv, ctx=ctx, lse=lse, klen=klen32, causal=causal, pa=pa, p=p, seed=seed, T=T, B=B,
                 dist_penalty=dist_penalty)
        return y.view(T, B, D), c

    def self_attn_block_bwd(self, pfx, c, dy, d=None, nxt=None):
        """dy [T*B, D] gradient w.r.t. the block output; returns gradient w.r.t. the block input.
        d: dropout(dy) with this block's mask when the producer of dy already wrote it; nxt: (p, seed) of the dropout that
        consumes the result next -> returns (dx, dropout(dx)) from the LayerNorm-backward pass."""
        hp = self.hp
        T, B, D = c["T"], c["B"], self.hp.D
        if d is None:
            d = K.dropout(dy, c["p"], c["seed"] + 2) if c["p"] > 0 else dy
        dctx = self.linear_bwd(d, c["ctx"].view(T * B, D), pfx + "self_attn.out_proj")
        dqkv = torch.empty_like(c["qkv"])
        qkv = c["qkv"]
        K.attn_bwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], c["ctx"], dctx.view(T, B, D), c["lse"], hp.heads,
                   dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], klen=c["klen"], causal=c["causal"],
                   p_drop=c["pa"], seed=c["seed"] + 1, dist_penalty=c["dist_penalty"])
        dh = self.linear_bwd(dqkv.view(T * B, 3 * D), c["h"], pfx + "self_attn.qkv")
        return K.layernorm_bwd(dh, c["x"], c["mean"], c["rstd"], self.P(pfx + "self_attn_layer_norm.weight"),
                               self.G(pfx + "self_attn_layer_norm.weight"), self.G(pfx + "self_attn_layer_norm.bias"), dres=dy, drop=nxt)

    def cross_attn_block_fwd(self, pfx, x, enc2d, Ts, enc_klen32, training, seed):
        """decoder encoder-attention (transformer_layer.py:324-352): q from x, k/v from the encoder output."""
        hp = self.hp
        T, B, D = x.shape
        x2 = x.view(T * B, D)
        h, mean, rstd = K.layernorm_fwd(x2, self.P(pfx + "encoder_attn_layer_norm.weight"), self.P(pfx + "encoder_attn_layer_norm.bias"), hp.ln_eps)
        q = self.linear(h, pfx + "encoder_attn.q_proj").view(T, B, D)
        kv = self.linear(enc2d, pfx + "encoder_attn.kv").view(Ts, B, 2 * D)
        pa = hp.attention_dropout if training else 0.0
        ctx, lse = K.attn_fwd(q, kv[:, :, :D], kv[:, :, D:], hp.heads, klen=enc_klen32, causal=False, p_drop=pa, seed=seed + 1)
        p = hp.dropout if training else 0.0
        y = self.linear(ctx.view(T * B, D), pfx + "encoder_attn.out_proj", residual=x2, p_drop=p, seed=seed + 2)
        c = dict(x=x2, h=h, mean=mean, rstd=rstd, q=q, kv=kv, ctx=ctx, lse=lse, klen=enc_klen32, pa=pa, p=p, seed=seed, T=T, B=B, Ts=Ts, enc2d=enc2d)
        return y.view(T, B, D), c

    def cross_attn_block_bwd(self, pfx, c, dy, denc, d=None, nxt=None, accumulate=True):
        """accumulates (accumulate=False: writes) the encoder-output gradient into denc [Ts*B, D]; returns dx (d / nxt as in
        self_attn_block_bwd)."""
        hp = self.hp
        T, B, D, Ts = c["T"], c["B"], self.hp.D, c["Ts"]
        if d is None:
            d = K.dropout(dy, c["p"], c["seed"] + 2) if c["p"] > 0 else dy
        dctx = self.linear_bwd(d, c["ctx"].view(T * B, D), pfx + "encoder_attn.out_proj")
        dq = torch.empty_like(c["q"]); dkv = torch.empty_like(c["kv"])
        kv = c["kv"]
        K.attn_bwd(c["q"], kv[:, :, :D], kv[:, :, D:], c["ctx"], dctx.view(T, B, D), c["lse"], hp.heads,
                   dq, dkv[:, :, :D], dkv[:, :, D:], klen=c["klen"], causal=False, p_drop=c["pa"], seed=c["seed"] + 1)
        self.linear_bwd(dkv.view(Ts * B, 2 * D), c["enc2d"], pfx + "encoder_attn.kv", dx_out=denc, dx_accumulate=accumulate)
        dh = self.linear_bwd(dq.view(T * B, D), c["h"], pfx + "encoder_attn.q_proj")
        return K.layernorm_bwd(dh, c["x"], c["mean"], c["rstd"], self.P(pfx + "encoder_attn_layer_norm.weight"),
                               self.G(pfx + "encoder_attn_layer_norm.weight"), self.G(pfx + "encoder_attn_layer_norm.bias"), dres=dy, drop=nxt)

    def ffn_block_fwd(self, pfx, x, training, seed):
        """x + dropout(fc2(dropout_act(act(fc1(LN(x))))))   (transformer_layer.py:128-136)"""
        hp = self.hp
        T, B, D = x.shape
        x2 = x.view(T * B, D)
        h, mean, rstd = K.layernorm_fwd(x2, self.P(pfx + "final_layer_norm.weight"), self.P(pfx + "final_layer_norm.bias"), hp.ln_eps)
        pact = hp.activation_dropout if training else 0.0
        pre = torch.empty((T * B, hp.ffn), dtype=self.dtype, device=self.dev) if hp.act == "gelu" else None
        # ReLU in training: the backward needs one bit per activation ("stored value > 0" = active and kept), written by fc1's epilogue
        # in its own tile order when the product runs on the 256-wide kernel (K.relu_mask_bytes > 0), instead of a re-read of `a`
        nmask = K.relu_mask_bytes(T * B, hp.ffn, D) if (training and hp.act == "relu" and self.dtype == torch.bfloat16) else 0
        amask = torch.empty((nmask,), dtype=torch.uint8, device=self.dev) if nmask else None
        a = self.linear(h, pfx + "fc1", act=K.ACT_RELU_MASK if nmask else self.act_fwd, aux_out=amask if nmask else pre,
                        p_drop=pact, seed=seed + 3)
        p = hp.dropout if training else 0.0
        y = self.linear(a, pfx + "fc2", residual=x2, p_drop=p, seed=seed + 4)
        if self.relu_record is not None and hp.act == "relu" and pact == 0.0:
            self.relu_record[pfx + "ffn"] = (a > 0).view(T, B, hp.ffn)
        c = dict(x=x2, h=h, mean=mean, rstd=rstd, a=a, pre=pre, pact=pact, p=p, seed=seed, amask=amask)
        return y.view(T, B, D), c

    def ffn_block_bwd(self, pfx, c, dy, d=None, nxt=None):
        if d is None:
            d = K.dropout(dy, c["p"], c["seed"] + 4) if c["p"] > 0 else dy
        if self.hp.act == "relu":
            # a = relu(z) * keep/(1-p): a > 0 <=> active and kept; the 1/(1-p) factor goes in alpha
            if c.get("amask") is not None:
                da = self.linear_bwd(d, c["a"], pfx + "fc2", act=K.ACT_RELU_BWD_MASK, aux=c["amask"], alpha=1.0 / (1.0 - c["pact"]))
            else:
                da = self.linear_bwd(d, c["a"], pfx + "fc2", act=K.ACT_RELU_BWD, aux=c["a"], alpha=1.0 / (1.0 - c["pact"]))
        else:
            da = self.linear_bwd(d, c["a"], pfx + "fc2", act=K.ACT_GELU_BWD, aux=c["pre"])
            if c["pact"] > 0:
                K.dropout(da, c["pact"], c["seed"] + 3, out=da)
        dh = self.linear_bwd(da, c["h"], pfx + "fc1")
        return K.layernorm_bwd(dh, c["x"], c["mean"], c["rstd"], self.P(pfx + "final_layer_norm.weight"),
                               self.G(pfx + "final_layer_norm.weight"), self.G(pfx + "final_layer_norm.bias"), dres=dy, drop=nxt)

    # ------------------------------------------------------------------ one layer per C call (csrc/layer.hip)
    def _layer_desc(self, pfx, decoder, T, B, Ts, causal, dist_penalty):
        """ONE descriptor per layer: the shape-independent part (weight / gradient / LayerNorm pointers, rates) is filled once, the
        data-dependent part (T after CTC compression, target length, source length, mask flags) is patched in place per call --
        real batches have a new (T, B, Ts) almost every update, and a descriptor per shape would grow without bound and rebuild ~60
        pointer fields per miss.  Workspace sizes are remembered per shape in a small bounded table."""
        e = self._descs.get(pfx)
        if e is None:
            hp = self.hp
            d = L.LayerDesc(dtype=L.BF16, decoder=int(decoder), T=T, B=B, D=hp.D, heads=hp.heads, ffn=hp.ffn, Ts=Ts, gelu=int(hp.act == "gelu"),
                            causal=int(causal), dist_penalty=int(bool(dist_penalty)), ln_eps=hp.ln_eps, p_drop=hp.dropout,
                            p_attn=hp.attention_dropout, p_act=hp.activation_dropout)
            names = {"qkv": "self_attn.qkv", "o": "self_attn.out_proj", "fc1": "fc1", "fc2": "fc2"}
            lns = {"ln1": "self_attn_layer_norm", "ln2": "final_layer_norm"}
            if decoder:
                names.update(xq="encoder_attn.q_proj", xkv="encoder_attn.kv", xo="encoder_attn.out_proj")
                lns["lnx"] = "encoder_attn_layer_norm"
            for k, n in names.items():
                setattr(d, "w_" + k, self.W(pfx + n + ".weight").data_ptr()); setattr(d, "b_" + k, self.P(pfx + n + ".bias").data_ptr())
                setattr(d, "g_w_" + k, self.G(pfx + n + ".weight").data_ptr()); setattr(d, "g_b_" + k, self.G(pfx + n + ".bias").data_ptr())
            for k, n in lns.items():
                setattr(d, k + "_g", self.P(pfx + n + ".weight").data_ptr()); setattr(d, k + "_b", self.P(pfx + n + ".bias").data_ptr())
                setattr(d, "g_" + k + "_g", self.G(pfx + n + ".weight").data_ptr()); setattr(d, "g_" + k + "_b", self.G(pfx + n + ".bias").data_ptr())
            e = self._descs[pfx] = dict(desc=d, addr=ctypes.addressof(d), sizes={})
        d = e["desc"]
        d.T, d.B, d.Ts, d.causal, d.dist_penalty = T, B, Ts, int(causal), int(bool(dist_penalty))
        return e

    def _layer_sizes(self, e, key, training):
        """(workspace bytes for this mode, backward scratch bytes) of the descriptor's CURRENT shape"""
        k = key + (training, K.OPTION_EPOCH)             # the ReLU record's size follows the GEMM route options (s2t_gemm_relu_mask_bytes)
        sz = e["sizes"].get(k)
        if sz is None:
            if len(e["sizes"]) >= 64:
                e["sizes"].clear()
            sz = e["sizes"][k] = (K.layer_ws_bytes(e["desc"], training), K.layer_tmp_bytes(e["desc"]))
        return sz

    def layer_fwd(self, pfx, x, training, seeds, self_klen=None, causal=False, dist_penalty=False, enc2d=None, Ts=0, enc_klen=None):
        """x [T,B,D] -> the layer's output; seeds = (sa_attn, sa_out, xa_attn, xa_out, ffn_act, ffn_out)"""
        T, B, D = x.shape
        decoder = enc2d is not None
        shape = (T, B, Ts, bool(causal), bool(dist_penalty))
        e = self._layer_desc(pfx, decoder, *shape)
        nws, ntmp = self._layer_sizes(e, shape, bool(training))
        if nws == 0:
            raise RuntimeError("s2t_layer_ws_bytes refused the layer shape %r (D %d, ffn %d, heads %d)" % (shape, self.hp.D, self.hp.ffn, self.hp.heads))
        ws = torch.empty((nws,), dtype=torch.uint8, device=self.dev)
        y = torch.empty((T, B, D), dtype=self.dtype, device=self.dev)
        c = self._call
        c.training = int(training)
        c.self_klen = self_klen.data_ptr() if self_klen is not None else None
        c.enc_klen = enc_klen.data_ptr() if enc_klen is not None else None
        c.seed_sa_attn, c.seed_sa_out, c.seed_xa_attn, c.seed_xa_out, c.seed_ffn_act, c.seed_ffn_out = seeds
        c.x = x.data_ptr(); c.enc = enc2d.data_ptr() if decoder else None; c.y = y.data_ptr(); c.ws = ws.data_ptr()
        K.layer_fwd(e["addr"], ctypes.addressof(c))
        p = self.hp.dropout if training else 0.0
        return y, dict(composite=True, pfx=pfx, decoder=decoder, shape=shape, ntmp=ntmp, ws=ws, x=x, enc2d=enc2d, self_klen=self_klen,
                       enc_klen=enc_klen, seeds=seeds, training=training, p=p, T=T, B=B, Ts=Ts)

    def layer_bwd(self, cl, dy, dyd, nxt, denc=None, denc_accumulate=False):
        """dy [T*B, D] gradient w.r.t. the layer output (dyd: its dropout with the FFN's output mask, or None); nxt = (p, seed) of the
        dropout that consumes the result next.  Returns (dx, dropout(dx) or None); the weight-gradient products are appended to the
        engine's list for the grouped launch."""
        e = self._layer_desc(cl["pfx"], cl["decoder"], *cl["shape"])       # the layer's one descriptor, set back to this call's shape
        tmp = torch.empty((cl["ntmp"],), dtype=torch.uint8, device=self.dev)
        dx = torch.empty_like(dy)
        dxd = torch.empty_like(dy) if nxt is not None else None
        c = self._call
        c.training = int(cl["training"])
        sk, ek = cl["self_klen"], cl["enc_klen"]
        c.self_klen = sk.data_ptr() if sk is not None else None
        c.enc_klen = ek.data_ptr() if ek is not None else None
        c.seed_sa_attn, c.seed_sa_out, c.seed_xa_attn, c.seed_xa_out, c.seed_ffn_act, c.seed_ffn_out = cl["seeds"]
        c.x = cl["x"].data_ptr(); c.enc = cl["enc2d"].data_ptr() if cl["enc2d"] is not None else None; c.ws = cl["ws"].data_ptr()
        c.dy = dy.data_ptr(); c.dy_drop = dyd.data_ptr() if dyd is not None else None
        c.dx = dx.data_ptr(); c.dx_drop = dxd.data_ptr() if dxd is not None else None
        c.nxt_p, c.nxt_seed = nxt if nxt is not None else (0.0, 0)
        c.denc = denc.data_ptr() if denc is not None else None
        c.denc_accumulate = int(denc_accumulate)
        c.tmp = tmp.data_ptr()
        if self._n_items + 8 > K.WGRAD_GROUP_MAX:            # a decoder layer appends 7 products: never run out of room mid-layer
            K.wgrad_group_raw(self._n_items, ctypes.addressof(self._items))
            self._n_items = 0                                 # (their operands stay in _keep until the flush, which is harmless)
        c.items = ctypes.addressof(self._items); c.max_items = K.WGRAD_GROUP_MAX; c.n_items = self._n_items
        K.layer_bwd(e["addr"], ctypes.addressof(c))
        self._n_items = c.n_items
        self._keep.append((tmp, cl["ws"], cl["x"], cl["enc2d"], dy, dyd))
        return dx, dxd

    # ------------------------------------------------------------------ encoder
    def encoder_forward(self, src_tokens, src_lengths, training, seed=0, return_all_hiddens=False, keep=None):
        """Returns dict(out [T'',B,D], lengths int64 [B] (device), lengths_host list, ctc_out, ctc_lengths, ...), ctx.
        keep: LayerDrop decisions of this pass (--encoder-layerdrop, conv_transformer.py:238-243), one bool per layer, drawn by the
        caller on the host; a dropped layer is absent from both schedules (and, as in the reference, from `states`)."""
        hp = self.hp
        if keep is not None and hp.ctc_layer and not keep[hp.ctc_layer - 1]:
            # the reference leaves x_ctc / ctc_padding_mask unbound in that case and dies with UnboundLocalError at its return
            raise RuntimeError("LayerDrop removed encoder layer %d, the one --ctc-encoder-layer compresses after" % hp.ctc_layer)
        len_host = [int(v) for v in (src_lengths.tolist() if torch.is_tensor(src_lengths) else src_lengths)]
        # lengths after the two stride-2 convolutions: ceil(len / 2) twice (conv_transformer.py:213), on the host, one upload each for
        # the two integer widths the kernels read
        lens_host = [((l + 1) // 2 + 1) // 2 for l in len_host]
        len4 = torch.tensor(lens_host, dtype=torch.int64).to(self.dev, non_blocking=True)
        len4_32 = torch.tensor(lens_host, dtype=torch.int32).to(self.dev, non_blocking=True)
        x, sub = self.subsample_fwd(src_tokens, len4_32, training, seed * 1000)
        T4, B, D = x.shape
        klen = len4_32 if min(lens_host) < T4 else None             # create_mask -> None when nothing is padded
        ctx = dict(sub=sub, layers=[], ctc=None, T4=T4, B=B, state_layers=[])
        out = dict(ctc_out=None, ctc_lengths=None, ctc_lengths_host=None, pred=None, states=[] if return_all_hiddens else None)
        cur_len, cur_len_host, cur_klen = len4, lens_host, klen
        for l in range(hp.enc_layers):
            if keep is not None and not keep[l]:
                ctx["layers"].append(None)
                continue
            pfx = "encoder.layers.%d." % l
            s0 = seed * 1000 + 10 * (l + 1)
            if self.composite and self.defer_wgrad:
                x, cl = self.layer_fwd(pfx, x, training, (s0 + 1, s0 + 2, 0, 0, s0 + 3, s0 + 4), self_klen=cur_klen,
                                       dist_penalty=bool(hp.distance_penalty))
                ctx["layers"].append(cl)
            else:
                x, ca = self.self_attn_block_fwd(pfx, x, cur_klen, False, training, s0, dist_penalty=bool(hp.distance_penalty))
                x, cf = self.ffn_block_fwd(pfx, x, training, s0)
                ctx["layers"].append((ca, cf))
            if hp.ctc_layer == l + 1:
                Tn = x.shape[0]
                x2 = x.view(Tn * B, D)
                x_ctc = self.linear(x2, "encoder.ctc_fc", pad_rows=True).view(Tn, B, hp.V_src)   # row stride padded to 16 B
                pred, pmax, ctc_lse = K.ctc_argmax(x_ctc, want_lse=True)     # the CTC loss over x_ctc reuses the row log-sum-exps
                seg, rs, rl, new_len, w = K.ctc_rle(pred, pmax, cur_len, hp.ctc_strategy)
                # greedy path for the host-side UER (logging): copied to pinned memory on the stream BEFORE the one host sync of the
                # forward pass (the compressed lengths), which then covers both transfers
                pred_host = torch.empty(pred.shape, dtype=pred.dtype, pin_memory=True)
                pred_host.copy_(pred, non_blocking=True)
                new_len_host = new_len.tolist()
                out["pred_host"] = pred_host
                Tout = max(new_len_host)
                xc = K.ctc_compress_fwd(x, w, rs, rl, new_len, Tout)
                ctx["ctc"] = dict(layer=l, x=x2, seg=seg, w=w, Tn=Tn, Tout=Tout)
                out.update(ctc_out=x_ctc, ctc_lengths=cur_len, ctc_lengths_host=cur_len_host, ctc_klen=cur_klen, pred=pred, ctc_lse=ctc_lse)
                x = xc
                cur_len, cur_len_host = new_len, new_len_host
                cur_klen = new_len.to(torch.int32) if min(new_len_host) < Tout else None
            if return_all_hiddens:
                out["states"].append(x)
                ctx["state_layers"].append(l)
        Tn = x.shape[0]
        xn, mean, rstd = K.layernorm_fwd(x.view(Tn * B, D), self.P("encoder.layer_norm.weight"), self.P("encoder.layer_norm.bias"), hp.ln_eps)
        ctx["final"] = dict(x=x.view(Tn * B, D), mean=mean, rstd=rstd)
        if return_all_hiddens and out["states"]:
            out["states"][-1] = xn.view(Tn, B, D)
        out.update(out=xn.view(Tn, B, D), lengths=cur_len, lengths_host=cur_len_host, klen=cur_klen)
        return out, ctx

    def encoder_backward(self, ctx, d_out, d_ctc_out=None, d_states=None):
        """d_out [T'',B,D] gradient w.r.t. the (layer-normed) encoder output; d_ctc_out [T4,B,V_src] or None;
        d_states: optional {layer index: gradient} for encoder_states consumers (criterion-owned CTC head)."""
        hp = self.hp
        B, D = ctx["B"], hp.D
        f = ctx["final"]
        if d_out is None:
            d_out = torch.zeros_like(f["x"])
        d_out = d_out.reshape(-1, D).contiguous()
        kept = [l for l in range(hp.enc_layers) if ctx["layers"][l] is not None]         # LayerDrop: the layers that ran
        top = kept[-1] if kept else -1
        if d_states and top in d_states:
            d_out = K.add_inplace(d_states[top].reshape(-1, D).contiguous(), d_out.clone())

        def touched(l):
            """the gradient entering layer l's output is modified between the blocks (CTC tap / compression): no fused mask"""
            return l < 0 or bool(d_states and l in d_states and l != top) or \
                (ctx["ctc"] is not None and ctx["ctc"]["layer"] == l)

        def ffn_drop(l):
            if l < 0:
                return None
            cl = ctx["layers"][l]
            p, sd = (cl["p"], cl["seeds"][5]) if isinstance(cl, dict) else (cl[1]["p"], cl[1]["seed"] + 4)
            return (p, sd) if self.fuse_bwd_dropout and p > 0 and not touched(l) else None

        dxd = None
        nxt = ffn_drop(top)
        dx = K.layernorm_bwd(d_out.reshape(-1, D).contiguous(), f["x"], f["mean"], f["rstd"], self.P("encoder.layer_norm.weight"),
                             self.G("encoder.layer_norm.weight"), self.G("encoder.layer_norm.bias"), drop=nxt)
        if nxt is not None:
            dx, dxd = dx
        for l in reversed(range(hp.enc_layers)):
            pfx = "encoder.layers.%d." % l
            if ctx["layers"][l] is not None:
                below = max([k for k in kept if k < l], default=-1)          # the next layer down that ran
                if d_states and l in d_states and l != top:
                    K.add_inplace(d_states[l].reshape(-1, D).contiguous(), dx)
                if ctx["ctc"] is not None and ctx["ctc"]["layer"] == l:
                    cc = ctx["ctc"]
                    dxk = torch.empty((cc["Tn"], B, D), dtype=self.dtype, device=self.dev)
                    K.ctc_compress_bwd(dx.view(cc["Tout"], B, D), cc["w"], cc["seg"], dxk)
                    dxk = dxk.view(cc["Tn"] * B, D)
                    if d_ctc_out is not None:
                        self.linear_bwd(d_ctc_out.reshape(-1, hp.V_src), cc["x"], "encoder.ctc_fc", dx_out=dxk, dx_accumulate=True)
                    dx = dxk
                if isinstance(ctx["layers"][l], dict):
                    dx, dxd = self.layer_bwd(ctx["layers"][l], dx, dxd, ffn_drop(below))
                else:
                    ca, cf = ctx["layers"][l]
                    nxt = (ca["p"], ca["seed"] + 2) if self.fuse_bwd_dropout and ca["p"] > 0 else None
                    dx = self.ffn_block_bwd(pfx, cf, dx, d=dxd, nxt=nxt)
                    dx, dxd = dx if nxt is not None else (dx, None)
                    nxt = ffn_drop(below)
                    dx = self.self_attn_block_bwd(pfx, ca, dx, d=dxd, nxt=nxt)
                    dx, dxd = dx if nxt is not None else (dx, None)
            if l == hp.enc_layers - 1:
                self._ready("encoder.layer_norm.")
            if hp.ctc_layer == l + 1:
                self._ready("encoder.ctc_fc.")
            self._ready(pfx)
            if self.wgrad_flush_layers and l > 0 and l % self.wgrad_flush_layers == 0:
                self.flush_wgrad()          # data-parallel runs: the upper layers' gradients go to the reducer while the lower ones run
        self.subsample_bwd(ctx["sub"], dx)
        for n in ("encoder.fc3.", "encoder.attn_2d.", "encoder.bn.1.", "encoder.convolutions.1.", "encoder.bn.0.", "encoder.convolutions.0."):
            self._ready(n)
        self.flush_wgrad()

    # ------------------------------------------------------------------ decoder
    def decoder_forward(self, prev_tokens, enc_out, enc_klen32, training, seed=0, pfx="decoder.", keep=None, attn_layer=None, attn_heads=None):
        """keep: LayerDrop decisions (--decoder-layerdrop, fairseq/modules/layer_drop.py:11-44), one bool per layer, drawn by the caller.
        attn_layer: also return (ctx["attn"], f32 [B, L, Ts]) the encoder-attention probabilities of that layer averaged over its first
        attn_heads heads (fairseq/models/transformer.py:756-782: alignment_layer / alignment_heads); that layer then runs on the per-kernel
        schedule, whose q and k are at hand (bit-identical to the layer call: tests/test_engine_gpu.py)."""
        hp = self.hp
        B, L = prev_tokens.shape
        D = hp.D
        scale = 1.0 if hp.no_scale_embedding else math.sqrt(D)
        tok = prev_tokens.contiguous()
        x = K.embed_fwd(tok, self.W(pfx + "embed_tokens.weight"), self.table(hp.pad + 1 + L, hp.pad), scale, hp.pad)
        p = hp.dropout if training else 0.0
        lne = None
        if hp.layernorm_embedding:               # transformer.py:731-732: LayerNorm between the embedding sum and its dropout
            xn, m_, r_ = K.layernorm_fwd(x.view(L * B, D), self.P(pfx + "layernorm_embedding.weight"), self.P(pfx + "layernorm_embedding.bias"),
                                         hp.ln_eps)
            lne = dict(x=x.view(L * B, D), mean=m_, rstd=r_)
            x = xn.view(L, B, D)
        if p > 0:
            K.dropout(x, p, seed * 1000 + 501, out=x)
        tlen = tok.ne(hp.pad).sum(dim=1).to(torch.int32)            # integer bookkeeping (suffix padding)
        Ts = enc_out.shape[0]
        enc2d = enc_out.reshape(Ts * B, D)
        ctx = dict(tok=tok, layers=[], B=B, L=L, Ts=Ts, scale=scale, p=p, seed=seed, pfx=pfx, lne=lne)
        for l in range(hp.dec_layers):
            if keep is not None and not keep[l]:
                ctx["layers"].append(None)
                continue
            lp = pfx + "layers.%d." % l
            s = seed * 1000 + 510 + 10 * l
            if self.composite and self.defer_wgrad and l != attn_layer:
                x, cl = self.layer_fwd(lp, x, training, (s + 1, s + 2, s + 4, s + 5, s + 7, s + 8), self_klen=tlen, causal=True,
                                       enc2d=enc2d, Ts=Ts, enc_klen=enc_klen32)
                ctx["layers"].append(cl)
                continue
            x, c1 = self.self_attn_block_fwd(lp, x, tlen, True, training, s)
            x, c2 = self.cross_attn_block_fwd(lp, x, enc2d, Ts, enc_klen32, training, s + 3)
            if l == attn_layer:
                ctx["attn"] = K.attn_probs_avg(c2["q"], c2["kv"][:, :, :D], hp.heads, klen=enc_klen32, heads_used=attn_heads)
            x, c3 = self.ffn_block_fwd(lp, x, training, s + 4)
            ctx["layers"].append((c1, c2, c3))
        xn, mean, rstd = K.layernorm_fwd(x.view(L * B, D), self.P(pfx + "layer_norm.weight"), self.P(pfx + "layer_norm.bias"), hp.ln_eps)
        ctx["final"] = dict(x=x.view(L * B, D), xn=xn, mean=mean, rstd=rstd)
        logits = self.linear(xn, self.out_proj(pfx), bias=False, pad_rows=True)          # [L*B, V] time-major rows
        return logits, ctx

    # ------------------------------------------------------------------ incremental decoding (generation)
    def decoder_begin(self, enc_out, enc_klen32, max_steps, pfx="decoder."):
        """Per-hypothesis decoding state (fairseq incremental_state, multihead_attention.py:368-440 / transformer.py:690-760):
        the encoder-side K/V of every layer (computed once: the reference's static_kv) and a ring of fused q|k|v rows
        `[max_steps][N][3D]` per layer that the step GEMM writes in place (no concat, no copy)."""
        hp = self.hp
        Ts, N, D = enc_out.shape
        enc2d = enc_out.reshape(Ts * N, D)
        st = dict(pfx=pfx, N=N, Ts=Ts, klen=enc_klen32, kv_enc=[], cache=[], spare=[], steps=0, max_steps=max_steps,
                  attn_layer=None, attn_heads=None, attn=None)     # attn_layer set by the caller: decoder_step leaves st["attn"] [N, 1, Ts]
        for l in range(hp.dec_layers):
            lp = pfx + "layers.%d." % l
            st["kv_enc"].append(self.linear(enc2d, lp + "encoder_attn.kv").view(Ts, N, 2 * D))
            st["cache"].append(torch.empty((max_steps, N, 3 * D), dtype=self.dtype, device=self.dev))
            st["spare"].append(torch.empty((max_steps, N, 3 * D), dtype=self.dtype, device=self.dev))
        return st

    def decoder_reorder(self, st, order, encoder_side=False):
        """Beam re-ordering of the self-attention caches (transformer.py:840-852 reorder_incremental_state).  Pure data movement
        (gather of rows).  This package's generator never compacts sentences out of the batch and the hypotheses of one sentence
        share their encoder-side K/V, so for it those stay put (multihead_attention.py:385-386 makes the same shortcut).
        encoder_side=True is the general contract of the reference's interface: its own SequenceGenerator drops finished sentences
        (fairseq/sequence_generator.py:430-470), so the encoder-side K/V, the key lengths and the batch size follow `order` too."""
        n = st["steps"]
        newN = int(order.numel())
        if encoder_side:
            st["kv_enc"] = [kv.index_select(1, order) for kv in st["kv_enc"]]
            if st["klen"] is not None:
                st["klen"] = st["klen"].index_select(0, order)
        if newN != st["N"]:
            if not encoder_side:
                raise ValueError("decoder_reorder: %d hypotheses -> %d needs encoder_side=True" % (st["N"], newN))
            for l in range(len(st["cache"])):
                old = st["cache"][l]
                st["cache"][l] = torch.empty((st["max_steps"], newN, old.shape[2]), dtype=old.dtype, device=old.device)
                if n:
                    torch.index_select(old[:n], 1, order, out=st["cache"][l][:n])
                st["spare"][l] = torch.empty_like(st["cache"][l])
            st["N"] = newN
            return
        if n == 0:
            return
        for l in range(len(st["cache"])):
            torch.index_select(st["cache"][l][:n], 1, order, out=st["spare"][l][:n])
            st["cache"][l], st["spare"][l] = st["spare"][l], st["cache"][l]

    def decoder_step(self, st, last_tokens):
        """One decoding step for N hypotheses: last_tokens int64 [N] -> logits [N, V] (row stride padded).
        Eval mode (no dropout); the same kernels as training with Tq = 1."""
        hp = self.hp
        pfx, N, D, step = st["pfx"], st["N"], hp.D, st["steps"]
        assert step < st["max_steps"]
        scale = 1.0 if hp.no_scale_embedding else math.sqrt(D)
        x = K.embed_fwd(last_tokens.view(N, 1).contiguous(), self.W(pfx + "embed_tokens.weight"),
                        self.table(hp.pad + 2 + st["max_steps"], hp.pad), scale, hp.pad, pos_offset=step).view(N, D)
        if hp.layernorm_embedding:
            x, _, _ = K.layernorm_fwd(x, self.P(pfx + "layernorm_embedding.weight"), self.P(pfx + "layernorm_embedding.bias"), hp.ln_eps)
        for l in range(hp.dec_layers):
            lp = pfx + "layers.%d." % l
            h, _, _ = K.layernorm_fwd(x, self.P(lp + "self_attn_layer_norm.weight"), self.P(lp + "self_attn_layer_norm.bias"), hp.ln_eps)
            cache = st["cache"][l]
            K.gemm(h, self.W(lp + "self_attn.qkv.weight"), bias=self.P(lp + "self_attn.qkv.bias"), out=cache[step])
            ctx, _ = K.attn_fwd(cache[step:step + 1, :, :D], cache[:step + 1, :, D:2 * D], cache[:step + 1, :, 2 * D:], hp.heads)
            x = self.linear(ctx.view(N, D), lp + "self_attn.out_proj", residual=x)
            h, _, _ = K.layernorm_fwd(x, self.P(lp + "encoder_attn_layer_norm.weight"), self.P(lp + "encoder_attn_layer_norm.bias"), hp.ln_eps)
            q = self.linear(h, lp + "encoder_attn.q_proj").view(1, N, D)
            kv = st["kv_enc"][l]
            ctx, _ = K.attn_fwd(q, kv[:, :, :D], kv[:, :, D:], hp.heads, klen=st["klen"])
            if l == st.get("attn_layer"):
                st["attn"] = K.attn_probs_avg(q, kv[:, :, :D], hp.heads, klen=st["klen"], heads_used=st.get("attn_heads"))
            x = self.linear(ctx.view(N, D), lp + "encoder_attn.out_proj", residual=x)
            h, _, _ = K.layernorm_fwd(x, self.P(lp + "final_layer_norm.weight"), self.P(lp + "final_layer_norm.bias"), hp.ln_eps)
            a = self.linear(h, lp + "fc1", act=self.act_fwd)
            x = self.linear(a, lp + "fc2", residual=x)
        xn, _, _ = K.layernorm_fwd(x, self.P(pfx + "layer_norm.weight"), self.P(pfx + "layer_norm.bias"), hp.ln_eps)
        st["steps"] = step + 1
        return self.linear(xn, self.out_proj(pfx), bias=False, pad_rows=True)

    def decoder_backward(self, ctx, dlogits, denc=None):
        """dlogits [L*B, V] (time-major).  Returns the gradient w.r.t. the encoder output [Ts*B, D]
        (accumulated into `denc` when given: second decoder of the dual-decoder model)."""
        hp = self.hp
        pfx, B, L, D = ctx["pfx"], ctx["B"], ctx["L"], hp.D
        f = ctx["final"]
        dxn = self.linear_bwd(dlogits, f["xn"], self.out_proj(pfx), bias=False)
        def drop_of(c, off):
            return (c["p"], c["seed"] + off) if self.fuse_bwd_dropout and c["p"] > 0 else None

        def out_drop(l):
            """(p, seed) of the dropout on layer l's output (its FFN's), for whoever produces the gradient that enters it"""
            cl = ctx["layers"][l]
            if isinstance(cl, dict):
                return (cl["p"], cl["seeds"][5]) if self.fuse_bwd_dropout and cl["p"] > 0 else None
            return drop_of(cl[2], 4)

        kept = [l for l in range(hp.dec_layers) if ctx["layers"][l] is not None]         # LayerDrop: the layers that ran
        nxt = out_drop(kept[-1]) if kept else None
        dx = K.layernorm_bwd(dxn, f["x"], f["mean"], f["rstd"], self.P(pfx + "layer_norm.weight"),
                             self.G(pfx + "layer_norm.weight"), self.G(pfx + "layer_norm.bias"), drop=nxt)
        dx, dxd = dx if nxt is not None else (dx, None)
        fresh = denc is None and len(kept) > 0          # the top layer's K/V data gradient WRITES the buffer: no fill, no read of zeros
        if denc is None:
            denc = (torch.empty if fresh else torch.zeros)((ctx["Ts"] * B, D), dtype=self.dtype, device=self.dev)
        for l in reversed(range(hp.dec_layers)):
            lp = pfx + "layers.%d." % l
            if isinstance(ctx["layers"][l], dict):
                below = max([k for k in kept if k < l], default=-1)
                dx, dxd = self.layer_bwd(ctx["layers"][l], dx, dxd, out_drop(below) if below >= 0 else None, denc=denc,
                                         denc_accumulate=not fresh)
                fresh = False
            elif ctx["layers"][l] is not None:
                c1, c2, c3 = ctx["layers"][l]
                below = max([k for k in kept if k < l], default=-1)
                nxt = drop_of(c2, 2)
                dx = self.ffn_block_bwd(lp, c3, dx, d=dxd, nxt=nxt)
                dx, dxd = dx if nxt is not None else (dx, None)
                nxt = drop_of(c1, 2)
                dx = self.cross_attn_block_bwd(lp, c2, dx, denc, d=dxd, nxt=nxt, accumulate=not fresh)
                fresh = False
                dx, dxd = dx if nxt is not None else (dx, None)
                nxt = out_drop(below) if below >= 0 else None
                dx = self.self_attn_block_bwd(lp, c1, dx, d=dxd, nxt=nxt)
                dx, dxd = dx if nxt is not None else (dx, None)
            if l == hp.dec_layers - 1:
                self._ready(pfx + "output_projection.")
                self._ready(pfx + "layer_norm.")
            self._ready(lp)
        if ctx["p"] > 0:
            K.dropout(dx, ctx["p"], ctx["seed"] * 1000 + 501, out=dx)
        if ctx.get("lne") is not None:
            e = ctx["lne"]
            dx = K.layernorm_bwd(dx.reshape(L * B, D).contiguous(), e["x"], e["mean"], e["rstd"], self.P(pfx + "layernorm_embedding.weight"),
                                 self.G(pfx + "layernorm_embedding.weight"), self.G(pfx + "layernorm_embedding.bias"))
        K.embed_bwd(ctx["tok"], dx.view(L, B, D), self.G(pfx + "embed_tokens.weight"), ctx["scale"], hp.pad)
        self._ready(pfx + "embed_tokens.")
        if self.flush_decoder_wgrad:
            self.flush_wgrad()
        return denc
